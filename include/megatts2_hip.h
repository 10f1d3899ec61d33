/* libmegatts2_hip - C ABI of the MI355X (gfx950) Mega-TTS 2 synthesis path.
 *
 * The reference (LSimon95/megatts2) has no plugin / FFI layer: its drop-in boundary is the Python
 * object surface of `models/megatts2.py`.  Each entry point below replaces one
 * method of that surface; the Python mirror in megatts2_amd/ binds them with ctypes and keeps the
 * reference's class / method names and tensor layouts.
 *
 * Conventions
 *  - Every `const float*` / `float*` / `int64_t*` / `int32_t*` NOT marked (host) is a DEVICE pointer
 *    (PyTorch-ROCm `tensor.data_ptr()`), f32 / int64 / int32, contiguous, 16-byte aligned.
 *  - Batched tensors use the reference's padded batch-first layouts; per-utterance true lengths are
 *    passed as (host) int32 arrays.  Padding positions of outputs are written as zeros.
 *  - `stream` is a hipStream_t (pass `torch.cuda.current_stream().cuda_stream`); all work is
 *    enqueued on it.  Calls that must size their output (mt2_adm_infer -> durations) document
 *    their host synchronisation.
 *  - Return value: 0 = ok, < 0 = error; `mt2_last_error()` gives the message (thread-local).
 *    Nothing throws across the boundary.
 *  - A model handle's weights are immutable after `mt2_model_finalize`.  The handle owns ONE activation
 *    workspace and its internal streams: calls on one handle are serialised (a mutex makes concurrent host
 *    threads safe; a call on a different stream than the previous call first waits, on the device, for that
 *    call's end).  For concurrency use one handle per thread / stream.  Apart from the handle there is no
 *    mutable global state in the library.  Batch semantics: every utterance is computed exactly as if it
 *    were alone (the reference is batch-1).
 */
#ifndef MEGATTS2_HIP_H
#define MEGATTS2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mt2_model mt2_model;

/* Hyper-parameters = the `model:` sub-trees of the reference's configs/config_{gan,plm,adm}.yaml
 * (reference models/megatts2.py:87-104,184-191,278-286) + the HiFi-GAN V1 generator topology. */
typedef struct mt2_config {
    /* MRTE (modules/mrte.py:64-84) */
    int32_t mel_bins, mrte_hidden, mrte_kernel, mrte_stride, mrte_n_layer, mrte_n_stack, mrte_n_block;
    int32_t content_ff_dim, content_n_heads, content_n_layers, phone_vocab;
    /* VQ prosody encoder (modules/vqpe.py:14-26) */
    int32_t vq_mel_bins, vq_stride, vq_hidden, vq_kernel, vq_n_layers, vq_n_stacks, vq_n_blocks, vq_bins, vq_dim;
    /* mel decoder (models/megatts2.py:31-54) */
    int32_t dec_kernel, dec_hidden, dec_n_stack, dec_n_block;
    /* PLM (models/megatts2.py:121-146) */
    int32_t plm_layers, plm_heads, plm_vq_dim, plm_tc_dim, plm_bins;
    /* ADM (models/megatts2.py:202-231) */
    int32_t adm_layers, adm_heads, adm_emb_dim, adm_tc_dim, adm_tc_emb_dim;
    /* HiFi-GAN V1 generator (speechbrain hub model in the reference, models/megatts2.py:321-323) */
    int32_t hg_in_dim, hg_init_channels, hg_n_up, hg_up_rates[8], hg_up_kernels[8];
    int32_t hg_n_res, hg_res_kernels[4], hg_res_dilations[4][3];
    float hg_slope;
    int32_t max_positions;   /* rows of the sine positional tables (embedding.py:66 builds 4000) */
    /* speechbrain HifiganGenerator.inference(): the mel is replicate-padded by this many frames on both sides
     * before the generator (hub model: 5), so decode_batch returns (T + 2*pad)*hop samples; 0 = plain forward */
    int32_t hg_inference_padding;
    /* edge mode of the generator's "same" convolutions (conv_pre, every ResBlock conv, conv_post): 0 = zero padding
     * (torch nn.Conv1d / transformers.SpeechT5HifiGan), 1 = per-utterance REFLECT padding - speechbrain's
     * nnet.CNN.Conv1d(padding="same") default padding_mode, what the hub model of the reference
     * (models/megatts2.py:321-323) is trained and run with.  The transposed convolutions are unaffected. */
    int32_t hg_reflect_pad;
} mt2_config;

const char* mt2_last_error(void);
const char* mt2_version(void);
/* 0 when a gfx950 device is visible and usable, < 0 otherwise (never falls back to a CPU path). */
int mt2_device_check(void);

/* ---- model lifetime (replaces MegaG/MegaPLM/MegaADM.from_pretrained, models/megatts2.py:107-117,
 * 184-198, 278-292).  Tensors are pushed one by one under their reference state_dict names, prefixed
 * by "G." / "plm." / "adm." / "hifigan."; `data` is a HOST pointer to `numel` f32 values (copied).
 * Additional tensors "pe.mrte" [max_positions, hidden], "pe.adm", "pe.plm": the sine tables of
 * modules/embedding.py:68-92 already multiplied by `alpha` (computed by the host mirror exactly
 * as the reference does).  finalize() checks the inventory strictly (load_state_dict(strict=True)),
 * repacks weights into GEMM-ready layouts and uploads them. */
mt2_model* mt2_model_create(const mt2_config* cfg);
int mt2_model_load_tensor(mt2_model* m, const char* name, const float* data /*host*/, const int64_t* shape /*host*/,
                          int ndim);
int mt2_model_finalize(mt2_model* m);
void mt2_model_destroy(mt2_model* m);
/* bytes of device memory held (weights, workspace) */
int mt2_model_memory(const mt2_model* m, size_t* weight_bytes, size_t* workspace_bytes);
/* ---- activation workspace.  The handle owns a bump arena over persistent hipMalloc chunks, reset by every call.
 * mt2_workspace_query: upper bound (bytes) of what ONE mt2_synthesize_batch call of this geometry takes (every
 * utterance at the maxima; flags as for mt2_synthesize_batch) - so that a server can pre-size the arena with
 * mt2_workspace_reserve and no hipMalloc happens on the hot path.  mt2_workspace_high_water: most bytes any call
 * has had in use so far. */
int mt2_workspace_query(const mt2_model* m, int B, int Np_max, int Tp_max, int Tm_cap, int flags, size_t* bytes);
int mt2_workspace_reserve(mt2_model* m, size_t bytes);
int mt2_workspace_high_water(const mt2_model* m, size_t* bytes);

/* ---- MRTE.tc_latent(phone, mel) (modules/mrte.py:154-171)
 * phone int64 [B, Np_max], mel f32 [B, Tp_max, mel_bins] -> out f32 [B, Np_max, hidden]. */
int mt2_mrte_tc_latent(mt2_model* m, void* stream, const int64_t* phone, const int32_t* phone_lens /*host*/,
                       int Np_max, const float* mel, const int32_t* mel_lens /*host*/, int Tp_max, int B,
                       float* out);
/* the mel-encoder part alone (mrte.mel_encoder, modules/convnet.py:202-210): -> [B, Tc_max, hidden],
 * Tc = (T-1)/stride + 1 */
int mt2_mrte_mel_context(mt2_model* m, void* stream, const float* mel, const int32_t* mel_lens /*host*/,
                         int Tp_max, int B, float* out, int Tc_max);

/* ---- MegaADM.infer(tc_latents) (models/megatts2.py:257-275)
 * tc_latent f32 [B, Np_max, tc_dim] -> dur int32 [B, Np_max] (0 in padding); `dur_float` (optional, may
 * be NULL) receives the un-rounded predictions.  Asynchronous on `stream`. */
int mt2_adm_infer(mt2_model* m, void* stream, const float* tc_latent, const int32_t* lens /*host*/, int Np_max,
                  int B, int32_t* dur, float* dur_float);
/* The same loop started from a FORCED history (no reference counterpart; parity tests of long sequences): the
 * un-rounded predictions of the first P positions of every sequence are given (p_prefix f32 [B, P], device), the
 * loop continues at position P for `max_steps` positions (0 = to the end).  P = n-1, max_steps = 1 is exactly the
 * reference's step t = n-1 (models/megatts2.py:264-273) on that history. */
int mt2_adm_infer_forced(mt2_model* m, void* stream, const float* tc_latent, const int32_t* lens /*host*/,
                         int Np_max, int B, const float* p_prefix, int P, int max_steps, int32_t* dur,
                         float* dur_float);

/* ---- LengthRegulator.forward(x, duration_tokens) (modules/mrte.py:42-60)
 * x f32 [B, Np_max, D], dur int32 (HOST) [B, Np_max] -> out f32 [B, Tm_max, D] (zero rows beyond sum(dur_b)).
 * Durations are taken from the host because the output size depends on them (the reference also
 * moves them to the CPU, mrte.py:53). */
int mt2_length_regulate(mt2_model* m, void* stream, const float* x, const int32_t* dur /*host*/,
                        const int32_t* lens /*host*/, int Np_max, int D, int B, float* out, int Tm_max);

/* ---- F.max_pool1d(x.transpose(1,2), k, ceil_mode=True).transpose(1,2) (models/megatts2.py:357-358)
 * x f32 [B, T_max, D] -> out f32 [B, Tq_max, D], Tq = ceil(T/k). */
int mt2_max_pool_ceil(mt2_model* m, void* stream, const float* x, const int32_t* lens /*host*/, int T_max, int D,
                      int B, int k, float* out, int Tq_max);

/* ---- MegaPLM.infer(tc_latent) (models/megatts2.py:165-181)
 * cond f32 [B, Tq_max, tc_dim] -> codes int64 [B, Tq_max] (0 in padding); `last_logits` optional
 * f32 [B, Tq_max, bins] receives the logits of every step's last position. */
int mt2_plm_infer(mt2_model* m, void* stream, const float* cond, const int32_t* lens /*host*/, int Tq_max, int B,
                  int64_t* codes, float* last_logits);
/* Prompt-conditioned decoding in the layout the PLM is trained on (modules/datamodule.py:201-212: the prompt's
 * pooled tc_latents and VQ-PE prosody codes in FRONT of the target's, BOS first): cond f32 [B, P + Tq_max, tc_dim]
 * holds P prompt rows then the target rows, prefix_codes int64 [B, P] (device) the prompt's codes; the history
 * starts as [BOS, prefix...] and decoding continues at position P for `max_steps` positions (0 = all).  lens =
 * TARGET lengths; codes / last_logits receive the target positions only.  P = 0 is mt2_plm_infer. */
int mt2_plm_infer_prompted(mt2_model* m, void* stream, const float* cond, const int32_t* lens /*host*/, int Tq_max,
                           int B, const int64_t* prefix_codes, int P, int max_steps, int64_t* codes,
                           float* last_logits);

/* ---- generator.vqpe.vq.decode(codes) (modules/quantization/vq.py:109-113)
 * codes int64 [n_q=1, B, Tq_max] -> out f32 [B, vq_dim, Tq_max]. */
int mt2_vq_decode(mt2_model* m, void* stream, const int64_t* codes, int B, int Tq_max, float* out);

/* ---- EuclideanCodebook.quantize (modules/quantization/core_vq.py:175-183): L2-argmin
 * x f32 [M, vq_dim] -> idx int64 [M] (lowest index on ties). */
int mt2_vq_quantize(mt2_model* m, void* stream, const float* x, int M, int64_t* idx);

/* ---- VQProsodyEncoder.forward(mel) (modules/vqpe.py:50-62), eval mode
 * mel f32 [B, T_max, mel_bins_full] (only the first vq_mel_bins are read) ->
 * zq f32 [B, T_max, vq_dim], codes int64 [1, B, Tq_max], ze (optional) f32 [B, Tq_max, vq_dim]. */
int mt2_vqpe_forward(mt2_model* m, void* stream, const float* mel, const int32_t* lens /*host*/, int T_max,
                     int mel_ld, int B, float* zq, int64_t* codes, int Tq_max, float* ze);

/* ---- generator.decoder(x) = ConvNet.forward (modules/convnet.py:115-119)
 * x f32 [B, decoder_in, T_max] ("B D T") -> mel f32 [B, mel_bins, T_max]. */
int mt2_mel_decoder(mt2_model* m, void* stream, const float* x, const int32_t* lens /*host*/, int T_max, int B,
                    float* mel);

/* ---- hifi_gan.decode_batch(mel) (speechbrain; models/megatts2.py:370): HiFi-GAN V1 generator
 * mel f32 [B, in_dim, T_max] -> wav f32 [B, 1, hop*(T_max + 2*hg_inference_padding)]; utterance b holds
 * hop*(len_b + 2*pad) samples (its own first / last frame replicated `pad` times, as
 * HifiganGenerator.inference does for a batch-1 call), zeros beyond. */
int mt2_hifigan(mt2_model* m, void* stream, const float* mel, const int32_t* lens /*host*/, int T_max, int B,
                float* wav);

/* ---- extract_mel_spec(samples) (modules/tokenizer.py:107-125): speechbrain `mel_spectogram` = torchaudio
 * Spectrogram(n_fft, win_length, hop_length, power=1, center=True, pad_mode="reflect", periodic Hann) ->
 * MelScale(n_mels, f_min, f_max, norm="slaney", mel_scale="slaney") -> log(clamp(x, clip)).
 * wav f32 [B, L_max] (16 kHz mono in the reference) -> mel f32 [B, T_max, n_mels], T_b = 1 + L_b / hop.
 * The STFT is an implicit conv on the GEMM engine: frame t = n_fft/hop consecutive hop-sized blocks of the
 * reflect-padded signal against a windowed DFT basis [2*(n_fft/2+1), n_fft].  n_fft % hop == 0, hop % 4 == 0. */
typedef struct mt2_audio_config {
    int32_t sample_rate, n_fft, hop_length, win_length, n_mels;
    float f_min, f_max, clip;
} mt2_audio_config;
int mt2_mel_spectrogram(mt2_model* m, void* stream, const mt2_audio_config* ac, const float* wav,
                        const int32_t* lens /*host*/, int L_max, int B, float* mel, int T_max);

/* ---- the whole of Megatts.forward's no_grad block (models/megatts2.py:353-368 [+370]) for a batch,
 * activations staying in the packed internal layout between stages.
 *   forced_dur   (host, optional) int32 [B, Np_max]: replaces the ADM's integer durations AFTER the ADM
 *                has run (benchmarks on synthetic weights).  NULL: the ADM's own
 *                durations are copied to the host (one stream synchronisation, as in the reference).
 *   forced_codes (device, optional) int64 [B, Tq_cap]: replaces the PLM (config C2).
 *   flags: bit0 run the PLM (ignored when forced_codes given), bit1 run the vocoder, bit2 skip the ADM.
 * Outputs: mel f32 [B, Tm_cap, mel_bins] (time-major), mel_lens (host) int32 [B], optional
 * dur_out int32 [B, Np_max] (device), codes_out int64 [B, Tq_cap] (device),
 * wav f32 [B, hop*(Tm_cap + 2*hg_inference_padding)].
 * Tm_cap / Tq_cap are capacities; an utterance longer than Tm_cap is an error.
 * Phone ids and forced codes (the whole zero-padded [B, Tq_cap] tensor) are range-checked on the device before
 * use - one stream synchronisation at the start of the call; an id outside its table is an error (the reference
 * raises IndexError from nn.Embedding), never an out-of-bounds read. */
#define MT2_RUN_PLM 1
#define MT2_RUN_VOCODER 2
#define MT2_SKIP_ADM 4
#define MT2_PROMPT_VQPE 8   /* also run VQProsodyEncoder.forward (modules/vqpe.py:50-62) on the PROMPT mel, on an internal
                             * stream beside the ADM: prompt_codes int64 [B, ceil(Tp_max / vq_stride)] (device) receives
                             * its prosody codes (what a prompt-conditioned PLM or stage-2 extraction consume) */
int mt2_synthesize_batch(mt2_model* m, void* stream, const int64_t* phone, const int32_t* phone_lens /*host*/,
                         int Np_max, const float* prompt_mel, const int32_t* prompt_lens /*host*/, int Tp_max,
                         int B, const int32_t* forced_dur /*host*/, const int64_t* forced_codes, int Tq_cap,
                         int flags, float* mel, int Tm_cap, int32_t* mel_lens /*host*/, int32_t* dur_out,
                         int64_t* codes_out, float* wav, int64_t* prompt_codes);

/* ---- prompt-conditioned synthesis as ONE call: the layout the PLM is trained on
 * (modules/datamodule.py:161-177,196-212) at inference.  `prompt_phone` int64 [B, Npp_max] (device) / `prompt_phone_lens` (host) /
 * `prompt_dur` int32 [B, Npp_max] (HOST) are the prompt utterance's own phones and alignment (sum over an utterance = its
 * prompt frames; every prompt pools to the same ceil(frames / vq_stride) = P).  The prompt's tc_latents (its phones against
 * the prompt mel, length-regulated by its alignment, max-pooled by vq_stride) stand in front of the target's as the PLM's
 * conditioning, the prompt's VQ-PE codes (computed here, returned in prompt_codes int64 [B, ceil(Tp_max / vq_stride)], the first P
 * of each row valid) behind the BOS; the PLM decodes the target's codes greedily from there (models/megatts2.py:165-181 on
 * that layout), then :361-368 [+370] as in mt2_synthesize_batch.  One MRTE mel-encoder pass per call; flags: MT2_RUN_VOCODER.
 * forced_dur (host, optional) replaces the ADM's durations after the ADM has run.  Other arguments as mt2_synthesize_batch. */
int mt2_synthesize_prompt_conditioned(mt2_model* m, void* stream, const int64_t* phone, const int32_t* phone_lens /*host*/,
                                      int Np_max, const float* prompt_mel, const int32_t* prompt_lens /*host*/, int Tp_max, int B,
                                      const int64_t* prompt_phone, const int32_t* prompt_phone_lens /*host*/, int Npp_max,
                                      const int32_t* prompt_dur /*host*/, const int32_t* forced_dur /*host*/, int Tq_cap, int flags,
                                      float* mel, int Tm_cap, int32_t* mel_lens /*host*/, int32_t* dur_out, int64_t* codes_out,
                                      float* wav, int64_t* prompt_codes);

/* ---- tuning.  Every switch lives in the handle (no process-global state): two handles do not see each other's settings.  None
 * changes results beyond f32 summation order.  The 30 names of round 6 (default in parentheses; the options the A/B records closed
 * were retired with their code and are unknown names now):
 *   stream groups   "ar_groups" (2; 1..8: the sequences of an autoregressive run are dealt into that many independent kernel chains
 *                   on internal HIP streams that fork from and join back into `stream`), "adm_groups" / "plm_groups" (0: per-stage
 *                   override), "voc_streams" (3: the three ResBlocks of a HiFi-GAN MRF side by side);
 *   arithmetic      "x3h" (15: bit mask of the fp16-pipe three-product forms - 1 loader-wave GEMM tiles, 2 K-split tiles of the AR
 *                   steps, 4 window convolutions, 8 long-sequence attention; 0 = the bf16 six-product forms, what the range guard's
 *                   repeat runs), "x6_conv" (1), "x6_gemm" (1), "x6_ks" (4), "win_conv" (1), "splitk" (1: K slices reduced by the next
 *                   LayerNorm), "skinny_tm" (1: launches of at most "skinny_rows" = 64 rows on the tile-major weight-streaming
 *                   kernel incl. its LayerNorm prologue), "skinny_nw" (16), "skinny_pairs" (1), "ln_pairs_adm" (2: LayerNorm
 *                   statistics handed from GEMM to GEMM in the ADM's layers), "ln_pairs_maxm" (1280), "ldr_prio" (3: s_setprio of the
 *                   loader waves), "a_planes" (3: bit mask of the producers that hand an activation to an x3h GEMM as fp16 planes -
 *                   1 LayerNorm kernels, 2 ff.0's epilogue -> ff.3, 4 attention -> out-projection);
 *   thresholds      "t_x3h_128" (72), "t_x6_256" (160), "t_x6_128" (72), "t_ks4" (256), "t_ks2" (640), "t32" (64), "t32x32" (160)
 *                   (tile counts at which the tile choice changes), "attn_x6_min" (192), "attn_lds_min" (640) (queries per sequence);
 *   measurement     "force_gemm_config" (-1), "stage_markers" (0), "ldr64" (0).
 * Unknown names are an error. */
int mt2_set_option(mt2_model* m, const char* name, int value);
int mt2_get_option(mt2_model* m, const char* name, int* value);
int mt2_set_ar_groups(mt2_model* m, int groups);

/* ---- measurement support: time (ms, HIP events on `stream`) spent in each stage of the last
 * mt2_synthesize_batch when profiling was enabled with mt2_set_profiling(m, 1).
 * names: "mrte", "adm", "regulate", "plm", "decoder", "vocoder".  Returns the number of stages. */
int mt2_set_profiling(mt2_model* m, int enable);
int mt2_last_stage_ms(mt2_model* m, const char** names, float* ms, int cap);

/* ---- kernel-level entry points (parity tests and micro-benchmarks call the engine directly)
 * C[M,N] = act(conv/linear(X) + bias) * scale + R, see megatts2_amd/csrc/mt2_kernels.h GemmP. */
int mt2_op_gemm(void* stream, const float* X, int ldx, int Rx, const int32_t* rowbase, int a_mul, int shift0,
                int taps, int dil, int Cin, const float* W, int ldw, const float* bias, const float* R, int ldr,
                const int32_t* valid, float* C, int ldc, int M, int N, int pro_act, float pro_slope, int epi_act,
                float out_scale, int force_cfg);
/* Window convolution with the weights additionally given as three bf16 planes W3 [3][N][K] (device uint16; truncation
 * split: W = W3[0] + W3[1] + W3[2] exactly): the kernel may then run on the bf16 matrix pipe in the f32-equivalent
 * 6-product form (conv_win_x6_kernel; force_cfg 34 / 58 / 59 = 32 / 64 / 128 channels, or -1 for the automatic choice; a retired
 * configuration index answers hipErrorNotSupported). */
int mt2_op_gemm_x6(void* stream, const float* X, int ldx, int Rx, int shift0, int taps, int dil, int Cin, const float* W,
                   const void* W3, const float* bias, const float* R, int ldr, const int32_t* valid, float* C, int ldc,
                   int M, int N, int pro_act, float pro_slope, int epi_act, int force_cfg);
/* The same launch with the weights ALSO as two fp16 planes (hi, lo * 2^11) of the row-scaled matrix - Wh: chunk-interleaved,
 * [N][ceil32(K) / 32] blocks of 128 bytes = 32 hi then 32 lo values of one row and 32-k chunk, 128-byte aligned - and the inverse
 * power-of-two row scales wh_inv [N] (mt2_x3h_split): the Linear / Conv1d of modules/transformer.py:35-57,88-102 and
 * modules/convnet.py:23-31 on the fp16 matrix pipe in the f32-equivalent THREE-product form (csrc/gemm_x3h.hip; force_cfg 103 = the
 * 128x128 loader tile, 95 / 96 / 97 = the K-split tiles, 98 / 99 / 100 = window convolutions of 32 / 64 / 128 channels, or -1).  Test
 * conventions of this entry point: force_cfg + 1000 = the loaders' 64-bit address form, + 2000 = X holds fp16 planes written by a
 * producer kernel (mt2_op_layernorm with act + 100, or this entry with + 4000), + 4000 = C is stored as such planes (loader tile only).
 * range_flag (device int32, may be NULL): |= 1 when an activation with |a| >= 65504 was converted (fp16 range). */
int mt2_op_gemm_x3h(void* stream, const float* X, int ldx, int Rx, int shift0, int taps, int dil, int Cin, const float* W,
                    const void* W3, const void* Wh, const float* wh_inv, const float* bias, const float* R, int ldr,
                    const int32_t* valid, float* C, int ldc, int M, int N, int pro_act, float pro_slope, int epi_act, int force_cfg,
                    int32_t* range_flag);
/* host helper: the fp16-pipe operand format of a row-major f32 matrix W [rows][row_len] (host memory): planes
 * [rows][2 * ceil32(row_len)] uint16 (fp16 bit patterns; per row and 32-k chunk 32 hi values, then 32 lo values; K zero-padded to a
 * multiple of 32), inv [rows] */
int mt2_x3h_split(const float* W, long long rows, long long row_len, uint16_t* planes, float* inv);
/* Range guard of the fp16-pipe kernels inside a model handle (option "x3h", default 15): waits for the handle's last call, then
 * *tripped = 1 (and the guard is re-armed) when that call converted an activation outside the fp16 range - its outputs are then
 * to be discarded and the call repeated with mt2_set_option(m, "x3h", 0) (the bf16 six-product form has f32's exponent range). */
int mt2_x3h_guard(mt2_model* m, int* tripped);
/* LayerNorm statistics handed from GEMM to GEMM (round 5; the AR layers of models/megatts2.py:172-179,264-273 =
 * TransformerEncoderLayer.forward, modules/transformer.py:88-102: `x = x + out_proj(...)` followed by `norm2(x)`, `x = x + ff(...)`
 * followed by the next layer's `norm1(x)`).  ONE linear launch C = epi(X @ W^T + bias) + R on a bf16-pipe (x6) tile with
 *   stat_out != NULL: the epilogue also writes, per row and per wave tile of *stat_w columns, the pair (mean, M2) of the final C
 *     values: stat_out[M][*stat_nt][2] (*stat_nt = 0: the chosen tile has no such epilogue, nothing was written);
 *   ln_stat  != NULL: C = LayerNorm(X) @ Wo^T + b in its algebraic form on those pairs - W / W3 = gamma-folded weights
 *     W'[n,k] = gamma[k] Wo[n,k], bias = c[n] = sum_k beta[k] Wo[n,k] + b[n], ln_s[n] = sum_k W'[n,k]; mean / rstd of source row r
 *     merged (Chan, fixed order) from ln_stat[r][ln_nt][2] with ln_w columns per pair: rstd * (X W'^T - mean * s) + c. */
int mt2_op_gemm_x6_ln(void* stream, const float* X, int ldx, int Rx, int a_mul, int shift0, const float* W, const void* W3,
                      const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K, int epi_act,
                      int force_cfg, float* stat_out, int32_t* stat_nt, int32_t* stat_w, const float* ln_stat, int ln_nt,
                      int ln_w, const float* ln_s, float ln_eps);
/* Linear layers of at most 64 rows on a TILE-MAJOR copy of the weights (round 4; gemm_skinny_tm_kernel - what the AR steps of one
 * utterance and the last-row launches of every batched AR step run on: F.linear at models/megatts2.py:172-179,264-273 with a
 * handful of rows).  mt2_op_tile_major turns a row-major [N, K] matrix (N a multiple of 16, K of 64) into blocks of 16 columns x 64 k,
 * block (nb, kb) at (nb * K/64 + kb) * 1024 floats, inside a block [j][lane][i] = W[nb*16 + lane%16][kb*64 + j*16 + (lane/16)*4 + i].
 * mt2_op_gemm_tm: C[g][M,N] = epi(pro(X[g] rows m*a_mul + shift0) @ Wsub[g]^T + bias) with Wsub[g] = rows [n0, n0 + N), columns
 * [k0, k0 + K) of the whole [*, Kw] matrix, moved by g * w_gstride row-major elements per group (split-K slabs: w_gstride = K).
 * ln_gamma != NULL: LayerNorm(X rows; gamma, beta, eps) as the prologue (K <= 1024, groups = 1) instead of pro_act.
 * pro_act + 0x100: the eight-wave form of the kernel (default: sixteen / twelve waves split K at M <= 32, round 5). */
int mt2_op_tile_major(void* stream, const float* W, int N, int K, float* out);
int mt2_op_gemm_tm(void* stream, const float* X, long long x_gstride, int ldx, int Rx, int a_mul, int shift0, const float* Wtm,
                   int Kw, int n0, int k0, long long w_gstride, int groups, const float* bias, const float* R, int ldr,
                   const int32_t* valid, float* C, long long c_gstride, int ldc, int M, int N, int K, int pro_act, float pro_slope,
                   int epi_act, const float* ln_gamma, const float* ln_beta, float ln_eps);
/* The same hand-off at a handful of rows (tile-major weight-streaming kernel, M <= 64): stat_out != NULL - the epilogue also writes
 * (mean, M2) of every final row per 16-column block, stat_out[M][N / 16][2]; ln_stat != NULL (with ln_gamma / ln_beta) - the LayerNorm
 * prologue takes mean / rstd of source row r from ln_stat[r][ln_nt = K / 16][2] instead of reading the rows. */
int mt2_op_gemm_tm_pairs(void* stream, const float* X, int ldx, int Rx, int a_mul, int shift0, const float* Wtm, int Kw,
                         const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K, int epi_act,
                         const float* ln_gamma, const float* ln_beta, float ln_eps, float* stat_out, const float* ln_stat, int ln_nt);
int mt2_op_layernorm(void* stream, const float* x, int ldx, const float* gamma, const float* beta, const float* R1,
                     int ldr1, const int32_t* valid, float* out, int ldo, int M, int C, float eps, int act);
int mt2_op_attention(void* stream, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                     float* O, int ldo, const int32_t* q_start, const int32_t* q_len, const int32_t* kv_start,
                     const int32_t* kv_len, int B, int H, int D, int max_qlen, float scale);
/* ... with the kernel choice exposed: lds_min_qlen = first query count served by the LDS-tiled kernel (0 never, < 0 default),
 * lds_waves = its query tiles per workgroup (8, otherwise 4); x6_min_qlen = first query count served by the bf16-pipe
 * kernel (f32-equivalent six-product form, head dims 64 / 96; 0 never, < 0 default); lds_waves + 32: the register kernel instead of the
 * head-dim-split kernel that serves D = 64 / 96 with at most 128 keys; lds_waves + 64: O receives fp16 planes (the operand format the
 * out-projection's x3h GEMM takes as it is: per 32 columns 32 hi | 32 lo fp16, same bytes per row; ldo % 32 == 0, O on 128 bytes);
 * lds_waves + 128: the bf16-pipe kernel in its fp16-pipe form (two fp16 planes per operand, three products; Q / K / V beyond the fp16
 * range are not reported through this entry point - the model's calls raise the handle's range guard);
 * max_kvlen: longest key range of the launch (0 = unknown). */
int mt2_op_attention_tuned(void* stream, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                           float* O, int ldo, const int32_t* q_start, const int32_t* q_len, const int32_t* kv_start,
                           const int32_t* kv_len, int B, int H, int D, int max_qlen, float scale, int lds_min_qlen,
                           int lds_waves, int x6_min_qlen, int max_kvlen);
/* The long-sequence kernel in its fp16-pipe form (two fp16 planes per operand, three products; head dims 64 / 96) with the range guard
 * exposed: range_flag (device int32) |= 1 when a Q / K / V value at or beyond 65504 was converted.  What mt2_synthesize_batch runs for
 * sequences of attn_x6_min queries or more under option x3h & 8 (reference modules/transformer.py:52-57, F.scaled_dot_product_attention). */
int mt2_op_attention_x3h(void* stream, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                         float* O, int ldo, const int32_t* q_start, const int32_t* q_len, const int32_t* kv_start,
                         const int32_t* kv_len, int B, int H, int D, int max_qlen, float scale, int lds_waves, int max_kvlen,
                         int32_t* range_flag);
/* Launch trace of the GEMM/conv engine (measurement only): between begin and end every launch is
 * bracketed by HIP events on its own stream.  end() reports, per tile configuration, the number of
 * launches, the executed FLOPs (2*M*N*K*groups) and the summed kernel time in ms, plus a last entry named
 * "union" = length of the union of all launch intervals (launches on different internal streams overlap);
 * returns the number of entries written (<= cap). */
int mt2_gemm_trace_begin(mt2_model* m);
int mt2_gemm_trace_end(mt2_model* m, int cap, const char** names, int64_t* launches, double* flops, double* ms);
/* text table "config M N K groups launches ms tflops" of the traced launches grouped by shape, slowest first;
 * call BEFORE mt2_gemm_trace_end (which frees the records); returns the bytes written or -1 */
int mt2_gemm_trace_shapes(mt2_model* m, char* buf, int cap, int top);
int mt2_gemm_config_count(void);
const char* mt2_gemm_config_name(int idx);
/* time `iters` back-to-back launches of one GEMM / conv (taps, dilation) with HIP events on `stream`, cycling
 * through `w_copies` copies of the weight matrix (> 1: weights are not L2-resident from the previous launch);
 * flags bit0: leaky-ReLU prologue, bit1: bias + residual + row-mask epilogue, bit2: clock probe - one wave of the
 * loader-wave x6 kernel reads s_memtime and s_memrealtime around its K loop; `sustained_ghz` (nullable) receives the
 * shader clock that launch ran at (0 when the configuration has no probe), bit3: no stderr report; average ms */
int mt2_op_gemm_x3h_ln(void* stream, const float* X, int ldx, int Rx, int a_mul, int shift0, const float* W, const void* W3,
                       const void* Wh, const float* wh_inv, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N,
                       int K, int epi_act, int force_cfg, float* stat_out, int32_t* stat_nt, int32_t* stat_w, const float* ln_stat,
                       int ln_nt, int ln_w, const float* ln_s, float ln_eps);   /* mt2_op_gemm_x6_ln with the fp16 planes as well */
int mt2_bench_gemm(void* stream, int M, int N, int K, int taps, int dil, int flags, int force_cfg, int iters,
                   int w_copies, float* avg_ms, char* cfg_name, int cfg_name_cap, double* sustained_ghz);

#ifdef __cplusplus
}
#endif
#endif /* MEGATTS2_HIP_H */
