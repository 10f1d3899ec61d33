// Internal kernel-launcher interface of libmegatts2_hip (gfx950 only).
//
// Every activation lives in HBM as a time-major row matrix [rows, channels] (f32, channels
// contiguous).  A batch of utterances is a *row set*: utterance b owns rows [off_b, off_b+len_b),
// separated from its neighbours by >= G all-zero "gap" rows.  Conv taps that reach over an
// utterance edge therefore read zeros - exactly the per-utterance zero padding the reference's
// batch-1 convolutions see - and every kernel re-zeroes gap rows in its epilogue
// through the `valid` row mask.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <vector>

namespace mt2 {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a launcher that needs more than the
// default dynamic LDS sets it once per (kernel, device).  `done` is the launcher's own bit mask of devices already served
// (device ordinal mod 64; a race between two host threads only repeats the idempotent call).
inline hipError_t dyn_lds_once(std::atomic<unsigned long long>& done, const void* fn, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

constexpr int kInvalidRow = -(1 << 30);   // rowbase sentinel: "this A row is all zeros"

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3,
                 ACT_LOGCLAMP = 4 };   // epilogue only: log(max(v, pro_slope))  (dynamic range compression)

// C[g][m, n] = epi( sum_{tap, c} pro(X[g][src(m) + tap*dil, c]) * W[g][n, tap*Cin + c] )
//   src(m)  = rowbase ? rowbase[m] : m * a_mul + shift0           (rows outside [0, Rx) read as 0)
//   epi(v)  = mask_m * ( act(v + bias[n]) * out_scale + R[g][m, n] )
// Linear layers: taps = 1.  Conv1d(k, dilation d, "same" padding): taps = k, shift0 = -(k-1)/2*d.
// Strided conv / gathers: rowbase.  Groups (blockIdx.z) batch independent problems of one shape
// (the parallel branches of ConvNetDouble) - a stride of 0 shares the operand between groups.
struct GemmP {
    const float* X; long long strideX; int ldx; int Rx;
    const int* rowbase; int a_mul; int shift0; int taps; int dil; int Cin;
    const float* W; long long strideW; int ldw;
    unsigned long long* dbg = nullptr;   // MT2_PHASE_TIMING builds only: per-phase cycle sums of one wave (tools/x6_phase_timing.py)
    int epi_t4 = 1;             // 16-byte-store epilogue where the layout allows it (set by launch_gemm from EngineOpts::epi_t4)
    int sk_nw = 8;              // gemm_skinny_tm_kernel: waves per workgroup that split K (8; 16 = option skinny_nw, set by launch_gemm)
    int a_planes = 0;           // 1: X holds fp16 planes ([32 hi | 32 lo] per 32 k of a row, x3h_planes.h's block layout without a row scale)
                                // written by a producer kernel (LnP::out_planes): x3h loader / K-split tiles only, no prologue, Cin % 32 = 0
    int c_planes = 0;           // 1: C RECEIVES fp16 planes (the same block layout: the A operand of the next launch's a_planes) instead of f32 -
                                // x3h loader tile only (16-byte-store epilogue), N % 32 = ldc % 32 = 0, C on 128 bytes, no residual, groups = 1
    int ldr64 = 0;              // x3h loaders: 1 = the 64-bit global_load_lds form even where buffer loads would do (EngineOpts::ldr64)
    int ldr_prio = 0;           // s_setprio of the loader waves of the loader-wave kernels (set by launch_gemm from EngineOpts::ldr_prio)
    const void* W3 = nullptr;   // optional: the same weights as three bf16 planes (truncation split, exact sum), addressed like
    long long w3_plane = 0;     // W (same ldw / strideW, in bf16 elements), planes w3_plane elements apart (0: N * ldw) -
                                // lets launch_gemm run on the bf16 matrix pipe in the f32-equivalent 6-product form
    const void* Wh = nullptr;   // optional: the same weights as two fp16 planes (hi, lo * 2^11) of the ROW-SCALED matrix, chunk-interleaved
    long long wh_ldb = 0;       // (x3h_planes.h: row n = blocks of 128 B, block c = [hi | lo] of k = 32 c ..): Wh points at the block of
    long long wh_gstride = 0;   // W's first element, wh_ldb = BYTES between rows (0: 4 * ceil32(ldw)), wh_gstride = bytes between groups;
    const float* wh_inv = nullptr;   // wh_inv[n] = the inverse (an exact power of two) of the scale of weight row n, n = 0 being W's first
    long long wh_inv_stride = 0;     // row; group g: + g * wh_inv_stride.  Lets launch_gemm run on the fp16 matrix pipe in the f32-equivalent
                                     // 3-product form (gemm_x3h.hip)
    int* x3h_flag = nullptr;    // device word: |= 1 when an x3h launch converted an activation with |a| >= 65504 (fp16 range; set by launch_gemm
                                // from EngineOpts::x3h_flag - the caller repeats the call without x3h)
    const float* Wtm = nullptr; // optional: the WHOLE matrix W points into, as tile-major blocks of 16 columns x 64 k (gemm_skinny.hip;
    int tm_n0 = 0, tm_k0 = 0;   // model_load.hip TmRange); (tm_n0, tm_k0) = block coordinates of W's first element (row / 16, column /
    int tm_kb = 0;              // 64), tm_kb = column blocks per block row (ldw / 64).  Lets launch_gemm stream the weights of a
                                // launch with <= 64 rows in 1-KiB pieces
    const float* bias; long long strideB;
    const float* R; long long strideR; int ldr;
    const int* valid;
    float* C; long long strideC; int ldc;
    int M, N, K, groups;
    int pro_act; float pro_slope; int epi_act; float out_scale;   // pro_slope doubles as the epilogue parameter (ACT_LOGCLAMP)
    // pro_act == 3 (LayerNorm prologue, linear layers with K <= 1024 only): A rows are normalised on the fly,
    // C = LN(X; ln_g, ln_b, ln_eps) W^T ...  launch_gemm returns hipErrorNotSupported when the tile configuration
    // it would choose has no such variant (big tiles) - callers then run launch_layernorm + a plain GEMM.
    // pro_act == 4 (ALGEBRAIC LayerNorm, K <= 1024): W = gamma-scaled weights W', bias = c, ln_g = s (see EncLayerW):
    // C = rstd * (X W'^T - mean * s) + c = LN(X) W^T + b; statistics in the prologue, K loop untouched.
    const float* ln_g; const float* ln_b; float ln_eps;
    // ---- LayerNorm statistics handed from GEMM to GEMM (round 5: the AR layers' stand-alone LayerNorm launches).
    // Producer side (any pro_act; groups == 1): stat_out != nullptr asks the epilogue to also write, per output row m and
    // per wave tile of `stat_w` columns, the pair (mean_t, M2_t) of the FINAL values C[m, tile] (after bias, activation,
    // residual and mask) as stat_out[(m * stat_nt + t) * 2 + {0, 1}], stat_nt = N / stat_w.  launch_gemm fills stat_nt /
    // stat_w for the tile it chooses (x6 K-split tiles: 32 columns; 128x128 loader tile: 64) and reports them in
    // EngineOpts::last_stat_nt / last_stat_w - 0 when that tile has no such epilogue (the launch itself still succeeds).
    float* stat_out = nullptr; int stat_nt = 0, stat_w = 0;
    // Consumer side, pro_act == 5 (ALGEBRAIC LayerNorm on PAIR statistics): like pro_act == 4 (W = W', bias = c, ln_g = s), but
    // mean / rstd of source row r come from merging the ln_nt pairs ln_stat[(r * ln_nt + t) * 2 ...] of ln_w columns each
    // (Chan's formula, fixed order) - no pass over K.  x6 tiles 55 / 84 / 85 / 86 only; hipErrorNotSupported otherwise.
    const float* ln_stat = nullptr; int ln_nt = 0, ln_w = 0;
};
// Tuning / measurement switches of the engine.  They live in the model handle (mt2_model::opts) or in a local
// object of a kernel-level entry point - never in process globals: two handles (or two threads) do not see each
// other's settings.  The defaults are what the product path runs.
struct TraceRec { int cfg; double flops; hipEvent_t e0, e1; int M, N, K, groups; };
struct EngineOpts {
    int force_cfg = -1;                                    // >= 0: tile configuration index for every GEMM launch
    int t_ks4 = 256, t_ks2 = 640, t32 = 64, t32x32 = 160;  // tile-choice thresholds in tiles (tools/gemm_sweep.py; t32 / t32x32 re-swept in the model after
                                                           // the x3h K-split tiles changed their ranking: the 64x64 k2 tile takes over from the 32x64 k4 and
                                                           // 32x32 k8 tiles earlier - C2 -2.5 ... -3 %, C3 -0.5 %, profiles/r06_opts_ab_block5_ks_thresholds.txt)
    bool splitk = true;          // split-K through the LayerNorm in the AR layers
    int voc_streams = 3;         // resblock chains of a vocoder stage in flight (1 = serial)
    bool win_conv = true;        // window-convolution kernel for narrow square convs (Cin = Cout in {32, 64, 128})
    bool x6_conv = true;         // ... on the bf16 matrix pipe, f32-equivalent 3-way split (6 products), where W3 planes exist
    bool x6_gemm = true;         // the implicit GEMMs likewise (loader-wave and K-split tiles)
    bool x6_splitk = true;       // K slices (through the next LayerNorm) to give the N = d AR GEMMs enough x6 tiles
    int x3h = 15;                // f32-equivalent THREE-product form on the fp16 pipe (gemm_x3h.hip) wherever an x6 tile has one and the
                                 // weights come with fp16 planes (GemmP::Wh); bits: 1 the 128x128 loader tile, 2 the K-split tiles of the
                                 // AR steps, 4 the window convolutions, 8 the long-sequence attention kernel (AttnP::x3h: C5 -3.3 %,
                                 // profiles/r06_opts_ab_block6_attention_x3h.txt); 0: everything stays x6
    int* x3h_flag = nullptr;     // device word of the range guard (GemmP::x3h_flag); the model handle owns one
    int t_x3h_128 = 72;          // x3h: from this many 128x128 tiles on the loader tile instead of the K-split tiles (t_x6_128's role)
    int t_x6_256 = 160, t_x6_128 = 72;   // ... from this many 256x128 / 128x128 tiles on (profiles/r02_gemm_sweep_x6.txt)
    int x6_ks = 4;               // x6 arithmetic + eight loader waves for the AR steps' K-split tiles (gemm_x6_ks_kernel): 0 off;
                                 // 1, 3: the 32x64 k4 and 64x64 k2/k4 tiles (84, 85); 2, 4: + the 32x32 k8 tile (86); 5: the
                                 // 64x64 tile only.  Default 4: isolated launches +10..50 %
                                 // (profiles/r03_gemm_sweep_x6k.txt), C3 step -1.6 % (profiles/r03_ab_interleaved_v1.txt)
    bool epi_t4 = true;          // DPP-transposed 16-byte-store epilogue for wave tiles without epilogue prefetch
    int a_planes = 3;            // producers of fp16 planes for x3h GEMMs that take their A operand as planes (GemmP::a_planes; wherever
                                 // gemm_takes_planes() says so).  Bit 0: LayerNorm -> Linear pairs of the AR layers and LayerNorm -> Conv1d inside
                                 // the conv stacks (LnP::out_planes); bit 1: ff.0's epilogue stores relu(..) as planes for ff.3 (GemmP::c_planes);
                                 // bit 2: the attention kernels store their output as planes for the out-projection (AttnP::o_planes) -
                                 // bit-identical, measured neutral, off by default (profiles/r06_opts_ab_block4_planes_producers.txt)
    bool ldr64 = false;          // tests / measurement: the x3h loaders' 64-bit global_load_lds form (what operands of 2 GiB or more get)
    int ldr_prio = 3;            // issue priority (s_setprio 0..3) of the loader waves (gemm_x6_ldr / gemm_x6_ks / conv_win_x6 kernels)
    int skinny_rows = 64;        // linear layers with at most this many rows (<= 64) run on the weight-streaming kernel of
    int attn_lds_min = 640;      // attention: from this many queries per sequence on the LDS-tiled kernel (AttnP::lds_min_qlen; 0 never)
    int attn_x6_min = 192;       // attention on the bf16 pipe (f32-equivalent, AttnP::x6_min_qlen) from this many queries on; 0: never
                                 // (C5 step 5525 -> 5314 ms at 192, 5376 at 448: profiles/r03_opts_ab.txt)
    int attn_lds_waves = 0;      // ... its query tiles per workgroup (AttnP::lds_waves)
    int attn_ds = 1;             // short sequences on the AR heads: head dim split over the waves as well (AttnP::ds_short)
    bool skinny_tm = true;       // ... on the tile-major weight copy where one exists (gemm_skinny_tm_kernel; LayerNorm prologue included)
    int skinny_groups = 1;       // ... only for launches with at least this many GemmP groups (split-K slabs)
    int skinny_pairs = 1;        // launches of at most skinny_rows rows: the residual GEMM's epilogue leaves (mean, M2) pairs per 16-column block
                                 // and the LayerNorm prologue of the consuming launch merges them instead of re-reading all M x K rows
    int skinny_nw = 16;          // waves of the tile-major weight-streaming kernel that split K (8; 16: four per SIMD - twelve for K = 768 -
                                 // at M <= 32: C1 -1.8 %, profiles/r05_opts_ab.txt)
    bool markers = false;        // a named no-op kernel at every stage boundary: lets tools/pmc_stage_summary.py attribute
                                 // the rocprofv3 --pmc rows of one step to stages (measurement only)
    bool trace_on = false;       // HIP events around every GEMM launch (measurement only)
    std::vector<TraceRec> trace;
    const char* last_cfg = "";   // name of the tile configuration the last launch used
    int last_stat_nt = 0, last_stat_w = 0;   // GemmP::stat_out of the last launch: pairs per row / columns per pair (0: none written)
    int ln_pairs = 0;            // (the value in force for the stage that is running: ln_pairs_adm inside the ADM, 0 elsewhere)
    int ln_pairs_adm = 2;        // ADM layers with more than skinny_rows rows: the residual GEMMs (out-projection, ff.3) write row statistics
                                 // as (mean, M2) pairs per wave tile in their epilogue and LN1 -> QKV / LN2 -> ff.0 run as ONE pair-fed
                                 // algebraic-LayerNorm GEMM - no stand-alone LayerNorm launch (1: where the producer is not K-split anyway;
                                 // 2: also un-split the producers with K <= ln_pairs_maxk).  The ADM alone (d = 768: both residual GEMMs
                                 // have K <= 1024) is where the hand-off pays at B = 32: interleaved C3 -0.43 %, C2 -1.3 %; in the PLM and
                                 // globally it measured +0.65 .. +2.0 % and was retired (profiles/r05_opts_ab.txt)
    int ln_pairs_maxk = 1024;    // ... ln_pairs = 2: the longest K chain that is taken out of the K split
    int ln_pairs_maxm = 1280;    // ... only for launches of at most this many rows: beyond, the consumers run on the 256x128 tile
                                 // (no pair-fed form) and a LayerNorm launch is no longer a latency item (C5: 5342 vs 5210 ms without a cap,
                                 // +0.46 % for the ADM alone at 2048, +0.1 % at 1280)
};
hipError_t launch_gemm(const GemmP& p, hipStream_t s, EngineOpts* o = nullptr);
// would launch_gemm run this launch (planes attached, no prologue) on an x3h tile that takes its A operand as fp16 planes (a_planes)?
bool gemm_takes_planes(const GemmP& p, const EngineOpts& o);
// would launch_gemm run this launch on an x3h loader tile whose epilogue can store C as fp16 planes (GemmP::c_planes)?
bool gemm_writes_planes(const GemmP& p, const EngineOpts& o);
// gemm_x3h.hip: the kernels of the fp16-pipe form by tile id and variant (prologue none / relu / leaky relu / - / - / pair statistics);
// nullptr: no such variant
enum X3hTile : int { X3H_LDR_128x128 = 0, X3H_KS_32x64_K4, X3H_KS_64x64_K2, X3H_KS_32x32_K8, X3H_WIN_256x64, X3H_WIN_128x128, X3H_WIN_256x32, kX3hTiles };
typedef void (*X3hKernel)(GemmP);
X3hKernel x3h_kernel(int tile, int variant);
// gemm_skinny.hip: weight-streaming linear layer for M <= 64 rows (taps = 1, no rowbase, K a multiple of 32)
bool gemm_skinny_eligible(const GemmP& p, int max_rows);
hipError_t launch_gemm_skinny(const GemmP& p, hipStream_t s);
// ... on a tile-major weight copy (GemmP::Wtm), optionally with the LayerNorm prologue (pro_act == 3, groups == 1, K <= 1024)
bool gemm_skinny_tm_eligible(const GemmP& p, int max_rows);
hipError_t launch_gemm_skinny_tm(const GemmP& p, hipStream_t s);
hipError_t launch_tile_major(const float* W, int N, int K, float* out, hipStream_t s);   // row-major [N, K] -> tile-major blocks
int gemm_num_configs();
const char* gemm_config_name(int idx);
// per tile configuration: launches, executed FLOPs, summed ms; last entry "union" (see gemm_f32.hip); -1 on error
int gemm_trace_collect(EngineOpts& o, int cap, const char** names, int64_t* launches, double* flops, double* ms);
// text table "config M N K groups launches ms tflops" of the traced launches grouped by shape, slowest first (call
// BEFORE gemm_trace_collect, which frees the records); returns the number of bytes written (< cap)
int gemm_trace_shapes(EngineOpts& o, char* buf, int cap, int top);

// Row LayerNorm over C channels (biased variance, eps inside the sqrt), one wave per row:
//   out[m, :] = mask_m * ( act( LN(x[m, :]) * gamma[g] + beta[g] ) + R1[m, :] + R2[m, :] )
// rows_per_group > 0: gamma/beta of group g = m / rows_per_group start at g * C.
// valid_rows > 0: the mask index is m % valid_rows (one mask shared by all groups);
// r1_rows > 0: the R1 row is m % r1_rows (one residual input shared by all groups).
struct LnP {
    const float* x; int ldx;
    const float* gamma; const float* beta; int rows_per_group;
    const float* R1; int ldr1; int r1_rows; const float* R2; int ldr2;
    const int* valid; int valid_rows;
    float* out; int ldo;
    int M, C; float eps; int act;
    int out_planes = 0;         // 1: `out` receives fp16 planes ([32 hi | 32 lo] per 32 channels, the A operand of a GemmP::a_planes launch;
                                // same bytes and row stride as f32; C % 32 == 0) instead of f32; x3h_flag: the range guard's device word
    int* x3h_flag = nullptr;
};
hipError_t launch_layernorm(const LnP& p, hipStream_t s);

// Split-K consumer (rowops.hip): x_new = R + bias + sum_g parts[g] (g = 0..S-1, fixed order);
// xout = x_new (may alias R; nullable), hout = LayerNorm(x_new) * gamma + beta.  parts[g] = parts + g*pstride, [M, C] dense.
struct LnReduceP {
    const float* parts; long long pstride; int S;
    const float* bias; const float* R; int ldr;
    const float* gamma; const float* beta;
    float* xout; int ldx; float* hout; int ldh;
    int M, C; float eps;
    int h_planes = 0;           // 1: hout receives fp16 planes (LnP::out_planes); the one-row-per-workgroup kernel only (M <= 4096)
    int* x3h_flag = nullptr;
};
hipError_t launch_ln_reduce(const LnReduceP& p, hipStream_t s);

// Non-causal multi-head attention over per-utterance row ranges (flash-style, f32 MFMA).
//   Q rows of utterance b: [q_start[b], q_start[b]+q_len[b]) in Q (ld ldq), head h at column h*D.
//   K/V likewise with kv_start/kv_len.  If q_start == nullptr the batch is uniform:
//   start = b * u_qstride (q) / b * u_kvstride (kv), len = u_qlen / u_kvlen.
//   Output rows: o_start[b] + i (default q_start), uniform: b * u_ostride + i (default u_qstride).
struct AttnP {
    const float* Q; int ldq; const float* K; int ldk; const float* V; int ldv;
    float* O; int ldo;
    const int* q_start; const int* q_len; const int* kv_start; const int* kv_len; const int* o_start;
    int u_qstride, u_qlen, u_kvstride, u_kvlen, u_ostride;
    int B, H, D, max_qlen; float scale;
    int max_kvlen;   // ragged launches: longest key range (0 = unknown), sizes the split-KV width
    int lds_min_qlen = 640;   // from this many queries on (and D <= 128) the LDS-tiled kernel shares K / V tiles between 8 query
                              // tiles of a workgroup (attn_f32_lds_kernel); 0: never
    int x6_min_qlen = 0;      // from this many queries on (D = 64 / 96) the f32-equivalent bf16-pipe kernel (attn_x6_kernel); 0: never
    int lds_waves = 0;        // query tiles per workgroup of that kernel: 8, otherwise 4
    int ds_short = 1;         // D = 64 / 96 and at most 128 keys: key tiles x head-dim slices per workgroup (attn_f32_ds_kernel)
    int x3h = 0;              // 1: the long-sequence kernel in its fp16-pipe form (two planes, three products; range-guarded through x3h_flag)
    int o_planes = 0;         // 1: O receives fp16 planes (planes_store.h: the A operand of the out-projection's GemmP::a_planes launch; same
    int* x3h_flag = nullptr;  // bytes and row stride as f32; D % 32 == 0, ldo % 32 == 0, O on 128 bytes); x3h_flag: the range guard's device word
};
hipError_t launch_attention(const AttnP& p, hipStream_t s);

// ---- row utilities (rowops.hip) ------------------------------------------------------------------
// out[r, :] = table[ids[idmap[r]], :] + pe[pos[r], :]   (idmap[r] < 0 -> zero row)
hipError_t launch_embed_pe(const float* table, int C, const int64_t* ids, const int* idmap, const int* pos,
                           const float* pe, float* out, int ldo, int R, int vocab, hipStream_t s);
// out[r, 0:C] = src[map[r], 0:C] (map[r] < 0 -> zeros);  generic row gather
hipError_t launch_gather_rows(const float* src, int lds_, const int* map, float* out, int ldo, int C, int R,
                              hipStream_t s);
// out[r, :] = max_{i < cnt[r]} src[first[r] + i, 0:C]   (cnt[r] == 0 -> zeros): ceil-mode max-pool
hipError_t launch_pool_max(const float* src, int lds_, const int* first, const int* cnt, float* out, int ldo,
                           int C, int R, hipStream_t s);
// out[r, :] = sum_g x[g*strideG + r*ld + :]
hipError_t launch_sum_groups(const float* x, long long strideG, int groups, int ld, float* out, int ldo, int C,
                             int R, hipStream_t s);
// dst[r, c] = (a[r,c] + b[r,c] + d[r,c]) * scale  (HiFi-GAN MRF mean)
hipError_t launch_avg3(const float* a, const float* b, const float* d, float scale, float* out, long long n,
                       hipStream_t s);
// MRF mean of the last stage + leaky ReLU + conv_post (Cout = 1) + tanh in one pass; x1 = x2 = nullptr: x0 is the mean already
hipError_t launch_conv_post(const float* x0, const float* x1, const float* x2, float scale, long long R, int ch, int k,
                            const float* w, const float* bias, float slope, const int* valid, float* out, hipStream_t s);
// padded [B, Tmax, C] (or channel-major [B, C, Tmax] when cmajor) <-> packed rows; rowmap[r] = b*Tmax + t or -1
hipError_t launch_pack_rows(const float* src, int C, int Tmax, int cmajor, const int* rowmap, float* dst, int ldd,
                            int R, hipStream_t s);
hipError_t launch_unpack_rows(const float* src, int lds_, int C, int Tmax, int cmajor, const int* rowmap,
                              float* dst, int R, hipStream_t s);
// AR step input rows (models/megatts2.py:172-176,264-269): uniform length n = t+1, A active sequences
//   ADM: x[j*n+i] = [tc_emb[tc_row[j]+i, 0:Dc] , w_dt[0:De] * p[j*pstride + i]] + pe[i]
hipError_t launch_adm_step_input(const float* tc_emb, int ld_tc, const int* tc_row, const float* w_dt,
                                 const float* p, int pstride, const float* pe, float* x, int Dc, int De,
                                 int n, int A, hipStream_t s);
//   PLM: x[j*n+i] = [cond[cond_row[j]+i, 0:Dc], emb[codes[j*cstride+i], 0:De]] + pe[i]
hipError_t launch_plm_step_input(const float* cond, int ld_c, const int* cond_row, const float* emb,
                                 const int64_t* codes, int cstride, const float* pe, float* x, int Dc, int De,
                                 int n, int A, int emb_rows, hipStream_t s);
// ADM head: p[j*pstride + n] = dot(x[j*xn + xn-1, 0:D], w)    (predict_layer, last position only; xn = rows
// per sequence in x: n for the full step matrix, 1 for the last-row matrix of the last layer)
hipError_t launch_adm_predict(const float* x, int D, const float* w, float* p, int pstride, int n, int xn, int A,
                              hipStream_t s);
// dur[i] = clamp(trunc(p + 0.5), 1, 128)  (models/megatts2.py:275)
hipError_t launch_adm_finalize(const float* p, int pstride, const int* lens, const int* slot_b, int32_t* dur,
                               float* flt, int dstride, int A, int nmax, hipStream_t s);
// out[slot_b[j]*ostride + t] = codes[j*cstride + 1 + skip + t] for t < lens[j] - skip (skip = prompt prefix length)
hipError_t launch_plm_finalize(const int64_t* codes, int cstride, const int* lens, const int* slot_b, int64_t* out,
                               int ostride, int A, int nmax, int skip, hipStream_t s);
// AR history initialisation (one launch instead of a host-staged copy):
//   ADM  p[j*pstride + 0] = 0 (models/megatts2.py:262), p[j*pstride + 1 + i] = prefix[slot_b[j]*P + i], rest 0
//   PLM  codes[j*cstride + 0] = bos (:170), codes[j*cstride + 1 + i] = prefix[slot_b[j]*pstride + i], rest 0
hipError_t launch_adm_init_hist(float* p, int pstride, const float* prefix, int P, const int* slot_b, int A,
                                hipStream_t s);
hipError_t launch_plm_init_hist(int64_t* codes, int cstride, int64_t bos, const int64_t* prefix, int P, int pstride,
                                const int* slot_b, int A, hipStream_t s);
// flag |= bit when an id used by the call is outside [0, hi): ids[map[r]] for map[r] >= 0 (map == nullptr: ids[r])
hipError_t launch_check_ids(const int64_t* ids, const int* map, int R, long long hi, int* flag, int bit,
                            hipStream_t s);
// dst[j*dpitch + c] = src[j*spitch + c], c < width, j < rows (float4 granularity)
hipError_t launch_copy_2d(const float* src, long long spitch, float* dst, long long dpitch, long long width, int rows,
                          hipStream_t s);
hipError_t launch_scatter_i64(const int64_t* src, const int* map, int64_t* out, int R, hipStream_t s);
hipError_t launch_expand_mask(const int* in, int factor, int* out, long long n, hipStream_t s);
hipError_t launch_unpack_wav(const float* src, const long long* start, const long long* len, float* out,
                             long long out_stride, long long max_len, int B, hipStream_t s);
// row arg-max with lowest-index ties (torch.argmax): out[j*ostride + ooff] = argmax_n x[j, 0:N]
hipError_t launch_argmax_rows(const float* x, int ldx, int N, int64_t* out, int ostride, int ooff, int A,
                              hipStream_t s);
// EuclideanCodebook.quantize (core_vq.py:175-183) given xe = x @ E^T:
//   idx[m] = argmax_j -((|x_m|^2 - 2*xe[m,j]) + ee[j]), lowest index on ties
hipError_t launch_vq_argmin(const float* x, int ldx, int D, const float* xe, int ldxe, const float* ee, int N,
                            const int* valid, int64_t* idx, int M, hipStream_t s);
// ee[j] = sum_d E[j,d]^2
hipError_t launch_row_sqnorm(const float* E, int D, float* ee, int N, hipStream_t s);
// decoder input (models/megatts2.py:361-366): out[r] = [tc[tcmap[r]], E[codes[codemap[r]]]]
hipError_t launch_decoder_input(const float* tc, int ld_tc, const int* tcmap, const float* E, const int64_t* codes,
                                const int* codemap, float* out, int Dc, int Dq, int R, int bins, hipStream_t s);
// zq rows (modules/vqpe.py:59-61): out[r] = E[codes[codemap[r]]]
hipError_t launch_codebook_rows(const float* E, const int64_t* codes, const int* codemap, float* out, int ldo,
                                int Dq, int R, int bins, hipStream_t s);
// mel front-end helpers (rowops.hip)
hipError_t launch_reflect_pad_blocks(const float* wav, long long wstride, const int* blk_b, const int* blk_t,
                                     const int* len, int hop, int pad, float* out, int R, hipStream_t s);
hipError_t launch_magnitude(const float* spec, int lds_, int F, float* out, int ldo, int M, hipStream_t s);
hipError_t launch_tanh_col(const float* x, int ldx, float* out, long long n, hipStream_t s);
// reflect halo rows of every utterance (rows start[b]*scale .. +len[b]*scale) in front of a "same" convolution
hipError_t launch_fill_reflect(float* x, int ld, int C, const int* start, const int* len, int B, long long scale, int G,
                               hipStream_t s);
// no-op kernel named mt2::stage_marker_kernel<ID> (ID 0..15): stage boundary in a kernel trace
hipError_t launch_stage_marker(int id, hipStream_t s);

}  // namespace mt2
