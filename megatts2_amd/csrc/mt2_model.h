// Host-side model object of libmegatts2_hip: weight store (GEMM-ready layouts in HBM), activation
// workspace, row-set planning and the stage drivers that enqueue kernels on the caller's stream.
#pragma once
#include "../../include/megatts2_hip.h"
#include "mt2_kernels.h"

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace mt2 {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
void check_hip(hipError_t e, const char* what, const char* file, int line);
#define MT2_HIP(x) ::mt2::check_hip((x), #x, __FILE__, __LINE__)
#define MT2_REQUIRE(cond, msg)                                                                   \
    do {                                                                                         \
        if (!(cond)) throw ::mt2::Error(std::string(msg) + " [" #cond "] at " __FILE__ ":" +     \
                                        std::to_string(__LINE__));                               \
    } while (0)

// Bump allocator over hipMalloc'd chunks; reset() at the start of every API call.  Chunks persist, so
// after the first call of a given shape no allocation happens on the hot path.
class Arena {
  public:
    ~Arena();
    void reset();
    void* alloc(size_t bytes);
    template <class T> T* get(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
    size_t capacity() const;
    size_t high_water() const { return high_; }      // most bytes in use at once since construction
    void reserve(size_t bytes);                      // make sure one chunk of at least `bytes` exists

  private:
    struct Chunk { char* p; size_t size, used; };
    std::vector<Chunk> chunks_;
    size_t high_ = 0;
};

// Host staging for H2D copies of small plans: PINNED memory owned by the handle.  hipMemcpyAsync from a pageable
// std::vector that dies before the stream reaches the copy is a use-after-free waiting to happen; everything the
// hot path uploads is first copied here.  Two pools alternate between API calls; a pool is recycled only after
// the event recorded at the end of the call that last used it has completed (bounds host run-ahead to 2 calls).
class PinnedPool {
  public:
    ~PinnedPool();
    void* alloc(size_t bytes);
    void reset() { for (auto& c : chunks_) c.used = 0; }
    hipEvent_t done = nullptr;
    bool recorded = false;

  private:
    struct Chunk { char* p; size_t size, used; };
    std::vector<Chunk> chunks_;
};

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

// A batch of per-utterance row ranges with >= G zero gap rows around each (mt2_kernels.h header).
struct RowSet {
    int B = 0, G = 0, R = 0, maxlen = 0;
    std::vector<int> len, off;
    const int* d_valid = nullptr;   // [R] 1 = real row
    const int* d_start = nullptr;   // [B]
    const int* d_len = nullptr;     // [B]
};

// all small integer arrays of one API call, uploaded with ONE H2D copy
class IntPlan {
  public:
    int add(const std::vector<int>& v);
    int add_fill(size_t n, int value);
    std::vector<int>& host() { return h_; }
    void upload(Arena& a, PinnedPool& pin, hipStream_t s);
    const int* dev(int offset) const { return d_ + offset; }

  private:
    std::vector<int> h_;
    int* d_ = nullptr;
};

struct ConvW { float* w = nullptr; float* b = nullptr; int cout = 0, cin = 0, k = 0; };
// A GEMM weight buffer that also exists as three bf16 planes (truncation split, exact sum; model_load.hip): the stage
// drivers look a weight pointer up here and hand the planes to launch_gemm (GemmP::W3), which may then run the launch
// on the bf16 matrix pipe in the f32-equivalent 6-product form.
// ... and as two fp16 planes of the row-scaled matrix + the inverse row scales (x3h_planes.h; rows of row_len elements): GemmP::Wh /
// wh_inv, the 3-product form on the fp16 pipe.
struct PlaneRange { const float* base; size_t n; const uint16_t* p3; const uint16_t* ph = nullptr; const float* inv = nullptr; size_t row_len = 0; };
// A linear layer's [N, K] weight matrix (N a multiple of 16, K of 64) that also exists as tile-major blocks (gemm_skinny.hip):
// the matrices of the AR encoders and the PLM head - what a launch of at most 64 rows streams.
struct TmRange { const float* base; size_t n; int K; const float* tm; };
// grouped residual stack: entry (s, blk) holds `groups` consecutive [C, k*C] matrices
struct StackW {
    float *w = nullptr, *b = nullptr, *g = nullptr, *be = nullptr;
    int C = 0, k = 0, nstack = 0, nblock = 0, groups = 0;
    size_t idx(int s, int blk) const { return (size_t)(s * nblock + blk) * groups; }
};
struct EncLayerW {
    float *ln1g, *ln1b, *ln2g, *ln2b, *wqkv, *bqkv, *wo, *bo, *ff0w, *ff0b, *ff1w, *ff1b;
    // algebraic-LayerNorm operands of the Linear-FF (AR) encoders, prepared once at load (gemm_f32.hip, PRO_LNA):
    //   W'[n,k] = gamma[k] W[n,k],  s[n] = sum_k W'[n,k],  c[n] = sum_k beta[k] W[n,k] + b[n]
    // for LN1 -> QKV (wqkv_l, sqkv, cqkv) and LN2 -> ff.0 (ff0_l, sff0, cff0); nullptr for conv-FF encoders
    float *wqkv_l = nullptr, *sqkv = nullptr, *cqkv = nullptr, *ff0_l = nullptr, *sff0 = nullptr, *cff0 = nullptr;
};
struct EncW {
    std::vector<EncLayerW> layers;
    int d = 0, ff = 0, heads = 0;
    bool conv_ff = false;
};
struct UpW { float *wlo, *whi, *bias; int cin, cout, stride; };
struct ResW { ConvW c1[3], c2[3]; int k; int dil[3]; };

struct StageTimer {
    std::vector<std::string> names;
    std::vector<hipEvent_t> ev;
};

}  // namespace mt2

struct mt2_model {
    mt2_config cfg{};
    bool finalized = false;
    bool has_g = false, has_adm = false, has_plm = false;
    std::map<std::string, mt2::HostTensor> host;
    std::vector<void*> dev_allocs;
    std::vector<mt2::PlaneRange> planes;       // sorted by base after finalize
    std::vector<mt2::TmRange> tm;              // likewise
    size_t weight_bytes = 0;
    mt2::Arena ws;

    // MRTE
    float *phone_emb = nullptr, *pe_mrte = nullptr;
    mt2::ConvW mel_first, mel_mid, mel_last;
    mt2::StackW mel_s1, mel_s2;
    mt2::EncW phone_enc;
    float *x_wq, *x_bq, *x_wkv, *x_bkv, *x_wo, *x_bo, *x_ng, *x_nb;
    // VQ prosody encoder
    mt2::ConvW vq_first, vq_last;
    mt2::StackW vq_s1, vq_s2;
    float *codebook = nullptr, *codebook_sq = nullptr;
    // decoder
    mt2::ConvW dec_first, dec_last;
    mt2::StackW dec_stack;
    // ADM
    mt2::EncW adm_enc;
    float *adm_wdt, *adm_wtc, *adm_wpred, *pe_adm;
    // PLM
    mt2::EncW plm_enc;
    float *plm_emb, *plm_wpred, *pe_plm;
    // HiFi-GAN
    bool has_vocoder = false;
    mt2::ConvW hg_pre, hg_post;
    std::vector<mt2::UpW> hg_up;
    std::vector<mt2::ResW> hg_res;

    // one API call at a time per handle (the handle owns the activation arena): calls from several host threads
    // are serialised here; calls on different streams are ordered by `ev_call_end` (capi.inc, CallScope)
    std::mutex call_mutex;
    mt2::PinnedPool pin[2];
    int pin_idx = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev_call_end = nullptr;
    bool call_pending = false;
    mt2::PinnedPool& pinned() { return pin[pin_idx]; }

    mt2::EngineOpts opts;     // tuning / measurement switches of THIS handle (no process globals)

    // AR stream groups (model_stages.hip): sequences are split into `ar_groups` independent kernel chains
    int ar_groups = 2;
    int adm_groups = 0, plm_groups = 0;     // per-stage override of ar_groups (0: ar_groups)
    std::vector<hipStream_t> aux_streams;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    // dedicated stream of the optional prompt VQ-PE of mt2_synthesize_batch (runs beside the ADM)
    hipStream_t vq_stream = nullptr;
    hipEvent_t ev_vq_fork = nullptr, ev_vq_join = nullptr, ev_vq_t0 = nullptr, ev_vq_t1 = nullptr;
    // internal streams that were forked from the caller's stream during the current API call and may still hold work:
    // CallScope's destructor joins them back on EVERY exit path (an exception thrown between a fork and its join must not
    // leave kernels that read the arena - or write caller buffers - running behind the call)
    bool aux_forked = false, vq_forked = false;
    // range check of caller-supplied gather indices: runs on its own stream (it depends on the call's INPUTS only), so
    // the verdict is read at the end of the call without waiting for the call's own kernels (model_stages.hip, IdCheck)
    hipStream_t id_stream = nullptr;
    hipEvent_t ev_id_fork = nullptr, ev_id_done = nullptr;
    int* id_flag_dev = nullptr;
    int* id_flag_host = nullptr;       // hipHostMalloc
    bool id_open = false;              // checks enqueued, verdict not read yet
    // range guard of the fp16-pipe GEMMs (gemm_x3h.hip): a device word the kernels OR into when an activation leaves the fp16
    // range; every API call copies it to the pinned host word before its end event (CallScope), mt2_x3h_guard reads it there
    int* x3h_flag_dev = nullptr;
    int* x3h_flag_host = nullptr;      // hipHostMalloc

    // mel front-end constants for the last mt2_audio_config seen (windowed DFT basis, mel filterbank)
    mt2_audio_config fe_cfg{};
    float *fe_basis = nullptr, *fe_fb = nullptr;
    int fe_nfreq = 0, fe_nfreq_pad = 0;

    bool profiling = false;
    std::vector<std::string> stage_names;
    std::vector<float> stage_ms;
};
