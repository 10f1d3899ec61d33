// Weight intake for libmegatts2_hip: strict inventory check (the reference loads with
// load_state_dict(strict=True), models/megatts2.py:116,196,290), repacking into GEMM-ready layouts
// and upload to HBM.
//
// Layouts produced here
//   Conv1d   weight [Cout, Cin, k]  ->  [Cout, k*Cin]  (tap-major K: k runs of Cin contiguous values,
//            matching the time-major activation rows the implicit-GEMM kernel walks)
//   Linear   weight [out, in]       ->  as is ([N, K], K contiguous)
//   q/k/v    three [d, d]           ->  one [3d, d] (+ bias [3d]); cross-attention: k/v -> [2d, d]
//   residual stacks: the same (stack, block) tensor of all parallel branches of a ConvNetDouble
//            (modules/convnet.py:186-197) contiguous, so one grouped launch covers every branch
//   ConvTranspose1d weight [Cin, Cout, k=2s] (padding s/2) -> two phase matrices [s/2*Cout, 2*Cin]
//            (see hifigan stage in model_stages.hip)
#include "mt2_model.h"
#include "x3h_planes.h"

#include <algorithm>
#include <cstring>

namespace mt2 {

thread_local std::string g_last_error;

void check_hip(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess)
        throw Error(std::string("HIP error ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" +
                    std::to_string(line));
}

// ---------------------------------------------------------------------------------------------------
Arena::~Arena() {
    for (auto& c : chunks_) (void)hipFree(c.p);
}
void Arena::reset() {
    for (auto& c : chunks_) c.used = 0;
}
void Arena::reserve(size_t bytes) {
    for (auto& c : chunks_)
        if (c.size >= bytes) return;
    char* p = nullptr;
    MT2_HIP(hipMalloc(reinterpret_cast<void**>(&p), bytes));
    chunks_.push_back({p, bytes, 0});
}
size_t Arena::capacity() const {
    size_t t = 0;
    for (auto& c : chunks_) t += c.size;
    return t;
}
void* Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    for (auto& c : chunks_) {
        if (c.size - c.used >= bytes) {
            void* p = c.p + c.used;
            c.used += bytes;
            size_t in_use = 0;
            for (auto& d : chunks_) in_use += d.used;
            if (in_use > high_) high_ = in_use;
            return p;
        }
    }
    size_t sz = bytes > (size_t(256) << 20) ? bytes : (size_t(256) << 20);
    char* p = nullptr;
    MT2_HIP(hipMalloc(reinterpret_cast<void**>(&p), sz));
    chunks_.push_back({p, sz, bytes});
    size_t in_use = 0;
    for (auto& d : chunks_) in_use += d.used;
    if (in_use > high_) high_ = in_use;
    return p;
}

PinnedPool::~PinnedPool() {
    for (auto& c : chunks_) (void)hipHostFree(c.p);
    if (done) (void)hipEventDestroy(done);
}
void* PinnedPool::alloc(size_t bytes) {
    bytes = (bytes + 63) & ~size_t(63);
    if (bytes == 0) bytes = 64;
    for (auto& c : chunks_)
        if (c.size - c.used >= bytes) {
            void* p = c.p + c.used;
            c.used += bytes;
            return p;
        }
    const size_t sz = bytes > (size_t(4) << 20) ? bytes : (size_t(4) << 20);
    char* p = nullptr;
    MT2_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), sz, hipHostMallocDefault));
    chunks_.push_back({p, sz, bytes});
    return p;
}

int IntPlan::add(const std::vector<int>& v) {
    while (h_.size() & 3) h_.push_back(0);
    const int off = (int)h_.size();
    h_.insert(h_.end(), v.begin(), v.end());
    return off;
}
int IntPlan::add_fill(size_t n, int value) {
    while (h_.size() & 3) h_.push_back(0);
    const int off = (int)h_.size();
    h_.resize(h_.size() + n, value);
    return off;
}
void IntPlan::upload(Arena& a, PinnedPool& pin, hipStream_t s) {
    if (h_.empty()) h_.push_back(0);
    d_ = a.get<int>(h_.size());
    void* stage = pin.alloc(h_.size() * sizeof(int));     // outlives this object: recycled two API calls later
    std::memcpy(stage, h_.data(), h_.size() * sizeof(int));
    MT2_HIP(hipMemcpyAsync(d_, stage, h_.size() * sizeof(int), hipMemcpyHostToDevice, s));
}

// ---------------------------------------------------------------------------------------------------
namespace {

struct Loader {
    mt2_model& m;
    std::map<std::string, bool> used;

    const HostTensor& get(const std::string& name, std::vector<int64_t> shape) {
        auto it = m.host.find(name);
        if (it == m.host.end()) throw Error("missing tensor in state_dict: " + name);
        if (it->second.shape != shape) {
            std::string s = "shape mismatch for " + name + ": got [";
            for (auto d : it->second.shape) s += std::to_string(d) + ",";
            s += "] expected [";
            for (auto d : shape) s += std::to_string(d) + ",";
            throw Error(s + "]");
        }
        used[name] = true;
        return it->second;
    }
    bool has(const std::string& name) const { return m.host.count(name) != 0; }

    // row_len > 0: a GEMM / conv weight matrix with rows of row_len elements - also kept as three bf16 planes and as two fp16
    // planes of the row-scaled matrix (PlaneRange)
    float* upload(const std::vector<float>& v, size_t row_len = 0) {
        const bool planes = row_len > 0;
        float* d = nullptr;
        const size_t bytes = v.size() * sizeof(float);
        MT2_HIP(hipMalloc(reinterpret_cast<void**>(&d), bytes ? bytes : 4));
        if (bytes) MT2_HIP(hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
        m.dev_allocs.push_back(d);
        m.weight_bytes += bytes;
        if (planes && !v.empty()) {
            MT2_REQUIRE(v.size() % row_len == 0, "weight buffer is not a whole number of rows");
            PlaneRange pr{d, v.size(), upload_planes(v)};
            const size_t rows = v.size() / row_len;
            std::vector<uint16_t> ph(rows * 2 * x3h_padded_k(row_len));      // chunk-interleaved [hi | lo] blocks, K padded to whole chunks
            std::vector<float> inv(rows);
            x3h_split_rows(v.data(), rows, row_len, ph.data(), inv.data());
            void* dp = nullptr;
            MT2_HIP(hipMalloc(&dp, ph.size() * sizeof(uint16_t)));
            MT2_HIP(hipMemcpy(dp, ph.data(), ph.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            m.dev_allocs.push_back(dp);
            void* di = nullptr;
            MT2_HIP(hipMalloc(&di, inv.size() * sizeof(float)));
            MT2_HIP(hipMemcpy(di, inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice));
            m.dev_allocs.push_back(di);
            m.weight_bytes += ph.size() * sizeof(uint16_t) + inv.size() * sizeof(float);
            pr.ph = static_cast<const uint16_t*>(dp); pr.inv = static_cast<const float*>(di); pr.row_len = row_len;
            m.planes.push_back(pr);
        }
        return d;
    }
    // f32 -> three bf16 planes by truncation: v = p1 + p2 + p3 EXACTLY (3 x 8 significant bits); [3][n] uint16
    const uint16_t* upload_planes(const std::vector<float>& v) {
        const size_t n = v.size();
        std::vector<uint16_t> pl(3 * n);
        for (size_t i = 0; i < n; ++i) {
            float r = v[i];
            for (int k = 0; k < 3; ++k) {
                uint32_t bits;
                std::memcpy(&bits, &r, 4);
                bits &= 0xffff0000u;
                float top;
                std::memcpy(&top, &bits, 4);
                pl[k * n + i] = (uint16_t)(bits >> 16);
                r -= top;                       // exact
            }
        }
        void* d = nullptr;
        MT2_HIP(hipMalloc(&d, pl.size() * sizeof(uint16_t)));
        MT2_HIP(hipMemcpy(d, pl.data(), pl.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        m.dev_allocs.push_back(d);
        m.weight_bytes += pl.size() * sizeof(uint16_t);
        return static_cast<const uint16_t*>(d);
    }
    // the tile-major copy of a linear layer's [N, K] matrix already uploaded at `base` (layout: gemm_skinny.hip)
    void upload_tm(const float* base, const std::vector<float>& v, int N, int K) {
        if ((N & 15) || (K & 63) || v.size() != (size_t)N * K) return;
        std::vector<float> t(v.size());
        const int KB = K / 64;
        for (int nb = 0; nb < N / 16; ++nb)
            for (int kb = 0; kb < KB; ++kb) {
                float* blk = t.data() + ((size_t)nb * KB + kb) * 1024;
                for (int j = 0; j < 4; ++j)
                    for (int kq = 0; kq < 4; ++kq)
                        for (int c = 0; c < 16; ++c) {
                            const float* src = v.data() + (size_t)(nb * 16 + c) * K + kb * 64 + j * 16 + kq * 4;
                            float* dst = blk + j * 256 + (kq * 16 + c) * 4;
                            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                        }
            }
        m.tm.push_back({base, v.size(), K, upload(t)});
    }
    float* vec(const std::string& name, int64_t n) { return upload(get(name, {n}).data); }
    float* mat(const std::string& name, int64_t r, int64_t c) { return upload(get(name, {r, c}).data); }

    // [Cout, Cin, k] -> [Cout, k*Cin]
    static void pack_conv(const HostTensor& t, std::vector<float>& out) {
        const int64_t co = t.shape[0], ci = t.shape[1], k = t.shape[2];
        const size_t base = out.size();
        out.resize(base + (size_t)co * ci * k);
        float* o = out.data() + base;
        for (int64_t a = 0; a < co; ++a)
            for (int64_t c = 0; c < ci; ++c)
                for (int64_t tap = 0; tap < k; ++tap)
                    o[(a * k + tap) * ci + c] = t.data[(a * ci + c) * k + tap];
    }
    ConvW conv(const std::string& p, int cout, int cin, int k) {
        ConvW w;
        w.cout = cout; w.cin = cin; w.k = k;
        std::vector<float> packed;
        pack_conv(get(p + ".weight", {cout, cin, k}), packed);
        w.w = upload(packed, (size_t)k * cin);
        w.b = vec(p + ".bias", cout);
        return w;
    }
    float* wmat(const std::string& name, int64_t r, int64_t c) { return upload(get(name, {r, c}).data, (size_t)c); }
    // ResidualBlockStack of `groups` parallel branches; prefix(l) gives branch l's stack prefix
    template <class F> StackW stack(F prefix, int groups, int C, int k, int nstack, int nblock) {
        StackW s;
        s.C = C; s.k = k; s.nstack = nstack; s.nblock = nblock; s.groups = groups;
        std::vector<float> w, b, g, be;
        for (int st = 0; st < nstack; ++st)
            for (int blk = 0; blk < nblock; ++blk)
                for (int l = 0; l < groups; ++l) {
                    const std::string p = prefix(l) + ".conv_stacks." + std::to_string(st) + ".blocks." +
                                          std::to_string(blk);
                    pack_conv(get(p + ".conv.weight", {C, C, k}), w);
                    const auto& bb = get(p + ".conv.bias", {C}).data;
                    b.insert(b.end(), bb.begin(), bb.end());
                    const auto& gg = get(p + ".norm.weight", {C}).data;
                    g.insert(g.end(), gg.begin(), gg.end());
                    const auto& ee = get(p + ".norm.bias", {C}).data;
                    be.insert(be.end(), ee.begin(), ee.end());
                }
        s.w = upload(w, (size_t)C * k); s.b = upload(b); s.g = upload(g); s.be = upload(be);
        return s;
    }
    EncW encoder(const std::string& prefix, int layers, int d, int ff, int heads, bool conv_ff) {
        EncW e;
        e.d = d; e.ff = ff; e.heads = heads; e.conv_ff = conv_ff;
        for (int l = 0; l < layers; ++l) {
            const std::string p = prefix + "." + std::to_string(l);
            EncLayerW w{};
            w.ln1g = vec(p + ".norm1.weight", d); w.ln1b = vec(p + ".norm1.bias", d);
            w.ln2g = vec(p + ".norm2.weight", d); w.ln2b = vec(p + ".norm2.bias", d);
            std::vector<float> qkv, bqkv;
            for (const char* n : {"w_q", "w_k", "w_v"}) {
                const auto& t = get(p + ".attn." + n + ".weight", {d, d}).data;
                qkv.insert(qkv.end(), t.begin(), t.end());
                const auto& bb = get(p + ".attn." + n + ".bias", {d}).data;
                bqkv.insert(bqkv.end(), bb.begin(), bb.end());
            }
            w.wqkv = upload(qkv, (size_t)d); w.bqkv = upload(bqkv);
            w.wo = wmat(p + ".attn.out_proj.0.weight", d, d);
            w.bo = vec(p + ".attn.out_proj.0.bias", d);
            if (conv_ff) {
                std::vector<float> a, c;
                pack_conv(get(p + ".ff.0.weight", {ff, d, 5}), a);
                pack_conv(get(p + ".ff.2.weight", {d, ff, 5}), c);
                w.ff0w = upload(a, (size_t)5 * d); w.ff0b = vec(p + ".ff.0.bias", ff);
                w.ff1w = upload(c, (size_t)5 * ff); w.ff1b = vec(p + ".ff.2.bias", d);
            } else {
                w.ff0w = wmat(p + ".ff.0.weight", ff, d); w.ff0b = vec(p + ".ff.0.bias", ff);
                w.ff1w = wmat(p + ".ff.3.weight", d, ff); w.ff1b = vec(p + ".ff.3.bias", d);
                // the AR encoders' matrices as tile-major blocks too: their launches of at most 64 rows (one utterance's
                // steps, the last-row launches of every batched step) stream them in 1-KiB pieces
                upload_tm(w.wqkv, qkv, 3 * d, d);
                upload_tm(w.wo, get(p + ".attn.out_proj.0.weight", {d, d}).data, d, d);
                upload_tm(w.ff0w, get(p + ".ff.0.weight", {ff, d}).data, ff, d);
                upload_tm(w.ff1w, get(p + ".ff.3.weight", {d, ff}).data, d, ff);
                // algebraic-LayerNorm operands (EncLayerW): sums in double, stored as f32
                auto fold = [&](const std::vector<float>& W, const std::vector<float>& b, const std::vector<float>& gam,
                                const std::vector<float>& bet, int N, float*& Wl, float*& sv, float*& cv) {
                    std::vector<float> wl((size_t)N * d), s(N), c(N);
                    for (int n = 0; n < N; ++n) {
                        double ss = 0.0, cc = b[n];
                        for (int k = 0; k < d; ++k) {
                            const float v = W[(size_t)n * d + k] * gam[k];
                            wl[(size_t)n * d + k] = v;
                            ss += (double)v;
                            cc += (double)bet[k] * (double)W[(size_t)n * d + k];
                        }
                        s[n] = (float)ss;
                        c[n] = (float)cc;
                    }
                    Wl = upload(wl, (size_t)d); sv = upload(s); cv = upload(c);      // + bf16 planes: the pair-fed form runs on the x6 tiles
                };
                fold(qkv, bqkv, get(p + ".norm1.weight", {d}).data, get(p + ".norm1.bias", {d}).data, 3 * d, w.wqkv_l, w.sqkv,
                     w.cqkv);
                fold(get(p + ".ff.0.weight", {ff, d}).data, get(p + ".ff.0.bias", {ff}).data, get(p + ".norm2.weight", {d}).data,
                     get(p + ".norm2.bias", {d}).data, ff, w.ff0_l, w.sff0, w.cff0);
            }
            e.layers.push_back(w);
        }
        return e;
    }
};

}  // namespace

void finalize_model(mt2_model& m) {
    const mt2_config& c = m.cfg;
    Loader L{m, {}};
    const int H = c.mrte_hidden;
    MT2_REQUIRE(H % 32 == 0 && (H / c.content_n_heads) % 32 == 0, "MRTE head dims must be multiples of 32");
    MT2_REQUIRE(c.mel_bins % 4 == 0 && c.vq_mel_bins % 4 == 0, "mel bins must be multiples of 4");
    MT2_REQUIRE(H <= 1024 && c.vq_hidden <= 1024 && c.dec_hidden <= 1024, "LayerNorm width limit is 1024");

    // Components are optional as a whole (the reference's MegaG / MegaPLM / MegaADM are separate
    // checkpoints, models/megatts2.py:308-312); a component that is present is checked strictly.
    m.has_g = L.has("G.mrte.phone_embedding.word_embeddings.weight");
    m.has_adm = L.has("adm.predict_layer.weight");
    m.has_plm = L.has("plm.predict_layer.weight");
    MT2_REQUIRE(m.has_g || m.has_adm || m.has_plm || L.has("hifigan.conv_pre.weight"), "no tensors were loaded");
    if (m.has_g) {
    // ---- G.mrte
    m.phone_emb = L.mat("G.mrte.phone_embedding.word_embeddings.weight", c.phone_vocab, H);
    (void)L.get("G.mrte.phone_pos_embedding.alpha", {1});
    m.pe_mrte = L.mat("pe.mrte", c.max_positions, H);
    m.mel_first = L.conv("G.mrte.mel_encoder.first_layer", H, c.mel_bins, c.mrte_kernel);
    m.mel_mid = L.conv("G.mrte.mel_encoder_middle_layer", H, H, c.mrte_stride + 1);
    for (int l = 0; l < c.mrte_n_layer; ++l) {   // aliases of the shared middle conv (mrte.py:101-115)
        const std::string p = "G.mrte.mel_encoder.layers." + std::to_string(l) + ".middle_layer";
        const auto& w = L.get(p + ".weight", {H, H, c.mrte_stride + 1});
        const auto& b = L.get(p + ".bias", {H});
        MT2_REQUIRE(w.data == m.host["G.mrte.mel_encoder_middle_layer.weight"].data &&
                        b.data == m.host["G.mrte.mel_encoder_middle_layer.bias"].data,
                    "mel_encoder middle_layer aliases must share one tensor");
    }
    m.mel_last = L.conv("G.mrte.mel_encoder.last_layer", H, H, c.mrte_kernel);
    m.mel_s1 = L.stack([](int l) { return "G.mrte.mel_encoder.layers." + std::to_string(l) + ".conv_stack1"; },
                       c.mrte_n_layer, H, c.mrte_kernel, c.mrte_n_stack, c.mrte_n_block);
    m.mel_s2 = L.stack([](int l) { return "G.mrte.mel_encoder.layers." + std::to_string(l) + ".conv_stack2"; },
                       c.mrte_n_layer, H, c.mrte_kernel, c.mrte_n_stack, c.mrte_n_block);
    m.phone_enc = L.encoder("G.mrte.phone_encoder.layers", c.content_n_layers, H, c.content_ff_dim,
                            c.content_n_heads, true);
    m.x_wq = L.wmat("G.mrte.mha.w_q.weight", H, H);
    m.x_bq = L.vec("G.mrte.mha.w_q.bias", H);
    {
        std::vector<float> kv, bkv;
        for (const char* n : {"w_k", "w_v"}) {
            const auto& t = L.get(std::string("G.mrte.mha.") + n + ".weight", {H, H}).data;
            kv.insert(kv.end(), t.begin(), t.end());
            const auto& bb = L.get(std::string("G.mrte.mha.") + n + ".bias", {H}).data;
            bkv.insert(bkv.end(), bb.begin(), bb.end());
        }
        m.x_wkv = L.upload(kv, (size_t)H);
        m.x_bkv = L.upload(bkv);
    }
    m.x_wo = L.wmat("G.mrte.mha.out_proj.0.weight", H, H);
    m.x_bo = L.vec("G.mrte.mha.out_proj.0.bias", H);
    m.x_ng = L.vec("G.mrte.norm.weight", H);
    m.x_nb = L.vec("G.mrte.norm.bias", H);

    // ---- G.vqpe
    const int VC = c.vq_hidden;
    MT2_REQUIRE(VC % 4 == 0 && c.vq_dim % 4 == 0, "VQ-PE widths must be multiples of 4");
    m.vq_first = L.conv("G.vqpe.convnet.first_layer", VC, c.vq_mel_bins, c.vq_kernel);
    m.vq_s1 = L.stack([](int l) { return "G.vqpe.convnet.layers." + std::to_string(l) + ".conv_stack1"; },
                      c.vq_n_layers, VC, c.vq_kernel, c.vq_n_stacks, c.vq_n_blocks);
    m.vq_s2 = L.stack([](int l) { return "G.vqpe.convnet.layers." + std::to_string(l) + ".conv_stack2"; },
                      c.vq_n_layers, VC, c.vq_kernel, c.vq_n_stacks, c.vq_n_blocks);
    m.vq_last = L.conv("G.vqpe.convnet.last_layer", c.vq_dim, VC, c.vq_kernel);
    {
        const std::string q = "G.vqpe.vq.vq.layers.0._codebook";
        const auto& inited = L.get(q + ".inited", {1});
        // core_vq.py:141-149,210: inited == 0 would make the reference run k-means inside forward()
        MT2_REQUIRE(inited.data[0] != 0.0f, "codebook buffer `inited` is 0: checkpoint holds an untrained VQ");
        (void)L.get(q + ".cluster_size", {c.vq_bins});
        (void)L.get(q + ".embed_avg", {c.vq_bins, c.vq_dim});
        m.codebook = L.mat(q + ".embed", c.vq_bins, c.vq_dim);
        float* sq = nullptr;
        MT2_HIP(hipMalloc(reinterpret_cast<void**>(&sq), sizeof(float) * c.vq_bins));
        m.dev_allocs.push_back(sq);
        MT2_HIP(launch_row_sqnorm(m.codebook, c.vq_dim, sq, c.vq_bins, nullptr));
        MT2_HIP(hipDeviceSynchronize());
        m.codebook_sq = sq;
    }

    // ---- G.decoder
    const int DH = c.dec_hidden, DIN = H + c.vq_dim;
    m.dec_first = L.conv("G.decoder.first_layer", DH, DIN, c.dec_kernel);
    m.dec_stack = L.stack([](int) { return std::string("G.decoder.conv_stack"); }, 1, DH, c.dec_kernel,
                          c.dec_n_stack, c.dec_n_block);
    m.dec_last = L.conv("G.decoder.last_layer", c.mel_bins, DH, c.dec_kernel);
    }   // has_g

    // ---- ADM (models/megatts2.py:201-231)
    if (m.has_adm) {
        const int d = c.adm_emb_dim + c.adm_tc_emb_dim, ff = c.adm_emb_dim * 4;
        MT2_REQUIRE(d % c.adm_heads == 0 && (d / c.adm_heads) % 32 == 0 && d <= 1024, "ADM head dim");
        MT2_REQUIRE(c.adm_emb_dim % 4 == 0 && c.adm_tc_emb_dim % 4 == 0, "ADM embedding widths");
        m.adm_enc = L.encoder("adm.adm.layers", c.adm_layers, d, ff, c.adm_heads, false);
        m.adm_wdt = L.upload(L.get("adm.dt_linear_emb.weight", {c.adm_emb_dim, 1}).data);
        m.adm_wtc = L.wmat("adm.tc_linear_emb.weight", c.adm_tc_emb_dim, c.adm_tc_dim);
        m.adm_wpred = L.upload(L.get("adm.predict_layer.weight", {1, d}).data);
        (void)L.get("adm.pos_emb.alpha", {1});
        m.pe_adm = L.mat("pe.adm", c.max_positions, d);
    }
    // ---- PLM (models/megatts2.py:120-146)
    if (m.has_plm) {
        const int d = c.plm_vq_dim + c.plm_tc_dim;
        MT2_REQUIRE(d % c.plm_heads == 0 && (d / c.plm_heads) % 32 == 0 && d <= 1024, "PLM head dim");
        m.plm_enc = L.encoder("plm.plm.layers", c.plm_layers, d, d * 4, c.plm_heads, false);
        m.plm_wpred = L.mat("plm.predict_layer.weight", c.plm_bins, d);
        L.upload_tm(m.plm_wpred, L.get("plm.predict_layer.weight", {c.plm_bins, d}).data, c.plm_bins, d);
        m.plm_emb = L.mat("plm.pc_embedding.weight", c.plm_bins + 2, c.plm_vq_dim);
        (void)L.get("plm.pos.alpha", {1});
        m.pe_plm = L.mat("pe.plm", c.max_positions, d);
    }
    // ---- HiFi-GAN (optional: only when its tensors were pushed)
    m.has_vocoder = L.has("hifigan.conv_pre.weight");
    if (m.has_vocoder) {
        int ch = c.hg_init_channels;
        m.hg_pre = L.conv("hifigan.conv_pre", ch, c.hg_in_dim, 7);
        for (int i = 0; i < c.hg_n_up; ++i) {
            const int s = c.hg_up_rates[i], k = c.hg_up_kernels[i], co = ch / 2;
            MT2_REQUIRE(k == 2 * s && s % 2 == 0, "ConvTranspose1d is supported for kernel = 2*stride, even stride");
            MT2_REQUIRE(ch % 4 == 0 && co % 4 == 0, "HiFi-GAN channel counts must be multiples of 4");
            const std::string p = "hifigan.upsampler." + std::to_string(i);
            const auto& w = L.get(p + ".weight", {ch, co, k});     // [Cin, Cout, k]
            const auto& b = L.get(p + ".bias", {co});
            // out[q*s + phi] = sum over the two input rows that reach phase phi (padding s/2):
            //   phi <  s/2: rows (q-1, q) with taps (phi + s/2 + s, phi + s/2)
            //   phi >= s/2: rows (q, q+1) with taps (phi + s/2, phi - s/2)
            const int hs = s / 2;
            std::vector<float> lo((size_t)hs * co * 2 * ch), hi((size_t)hs * co * 2 * ch), bias((size_t)hs * co);
            auto W = [&](int ci, int o, int tap) { return w.data[((size_t)ci * co + o) * k + tap]; };
            for (int ph = 0; ph < hs; ++ph)
                for (int o = 0; o < co; ++o) {
                    float* rl = lo.data() + ((size_t)ph * co + o) * 2 * ch;
                    float* rh = hi.data() + ((size_t)ph * co + o) * 2 * ch;
                    for (int ci = 0; ci < ch; ++ci) {
                        rl[ci] = W(ci, o, ph + hs + s);      // row q-1
                        rl[ch + ci] = W(ci, o, ph + hs);     // row q
                        rh[ci] = W(ci, o, ph + s);           // phi = ph + hs, row q: tap phi + hs - ... = ph + s
                        rh[ch + ci] = W(ci, o, ph);          // row q+1: tap phi - hs = ph
                    }
                    bias[(size_t)ph * co + o] = b.data[o];
                }
            UpW u;
            u.cin = ch; u.cout = co; u.stride = s;
            u.wlo = L.upload(lo, (size_t)2 * ch); u.whi = L.upload(hi, (size_t)2 * ch); u.bias = L.upload(bias);
            m.hg_up.push_back(u);
            for (int j = 0; j < c.hg_n_res; ++j) {
                ResW r;
                r.k = c.hg_res_kernels[j];
                const std::string rp = "hifigan.resblocks." + std::to_string(i * c.hg_n_res + j);
                for (int n = 0; n < 3; ++n) {
                    r.dil[n] = c.hg_res_dilations[j][n];
                    r.c1[n] = L.conv(rp + ".convs1." + std::to_string(n), co, co, r.k);
                    r.c2[n] = L.conv(rp + ".convs2." + std::to_string(n), co, co, r.k);
                }
                m.hg_res.push_back(r);
            }
            ch = co;
        }
        m.hg_post = L.conv("hifigan.conv_post", 1, ch, 7);
    }

    // strict: nothing unexpected
    for (auto& kv : m.host)
        if (!L.used.count(kv.first)) throw Error("unexpected tensor in state_dict: " + kv.first);
    m.host.clear();
    std::sort(m.planes.begin(), m.planes.end(), [](const PlaneRange& a, const PlaneRange& b) { return a.base < b.base; });
    std::sort(m.tm.begin(), m.tm.end(), [](const TmRange& a, const TmRange& b) { return a.base < b.base; });
    m.finalized = true;
}

}  // namespace mt2
