/*
 * ttsk.h — C ABI of libttsk_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the tts-king hot path.
 *
 * The reference (diff7/tts-king) is pure Python/PyTorch and has no FFI of its own; its hot
 * path bottoms out in ATen op calls.  Every entry point below replaces one (or a fused group) of those
 * op sites; the site is cited as `reference: <file>:<lines>`.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - Plain C: raw device pointers, sizes, a hipStream_t passed as `void*`.  No torch types.
 *   - The caller owns every buffer (inputs, outputs, workspaces).  The library allocates nothing, keeps no
 *     mutable global state, never synchronises: all work is enqueued on `stream` and is graph-capturable.
 *   - Return value: 0 on success, negative TTSK_E* on a rejected call (nothing was launched);
 *     `ttsk_last_error()` returns a thread-local message.
 *   - bf16 = raw uint16 storage (round-to-nearest-even from fp32); activations are channels-last
 *     [row][channel] with row = batch * seg_len + position.
 */
#ifndef TTSK_H
#define TTSK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTSK_VERSION 1

#define TTSK_OK 0
#define TTSK_EINVAL (-1)
#define TTSK_ELAUNCH (-2)

int ttsk_version(void);
const char* ttsk_last_error(void);

/* ------------------------------------------------------------------------------------------------ GEMM
 * One tiled MFMA kernel family behind every contraction on the path:
 *   Linear fwd / dX / dW              reference: fs_two/transformer/SubLayers.py:41-63, fastspeech2.py:102
 *   attention Q·Kᵀ, P·V and backward   reference: fs_two/transformer/Modules.py:14-24
 *   Conv1d as implicit GEMM (fwd/dX/dW) reference: SubLayers.py:96 (k=9,1), model/modules.py:337-355 (k=3),
 *                                                  transformer/Layers.py:59-67 (k=5), hifi/models.py:88-95,186,198
 *   ConvTranspose1d, polyphase          reference: hifi/models.py:166-176,189
 *
 *   C[z][m][n] = epilogue( alpha * sum_{tap} sum_{k} A[z][m (+shift_tap)][tap? k] * B[z][n][tap*btap + k] )
 */
enum {
  TTSK_GEMM_A_TR      = 1 << 0,  /* A stored [k][m] (contraction index is the row)                        */
  TTSK_GEMM_B_TR      = 1 << 1,  /* B stored [k][n]                                                        */
  TTSK_GEMM_C_F32     = 1 << 2,  /* C is fp32 (default bf16)                                               */
  TTSK_GEMM_RELU      = 1 << 3,  /* v = max(v, 0) after bias/residual                                      */
  TTSK_GEMM_ADD_R     = 1 << 4,  /* v += R[m][n]                                                           */
  TTSK_GEMM_R_F32     = 1 << 5,  /* R is fp32 (default bf16)                                               */
  TTSK_GEMM_MASK_G    = 1 << 6,  /* v = (G[m][n] > 0) ? v : 0   (ReLU backward, G = saved activation bf16) */
  TTSK_GEMM_LRELU_IN  = 1 << 7,  /* A := leaky_relu(A, in_slope) while staging (conv-A mode, HiFi-GAN)     */
  TTSK_GEMM_TANH      = 1 << 8,  /* v = tanh(v) last                                                       */
  TTSK_GEMM_ACCUM_C   = 1 << 9,  /* C_F32 only: C += v  (plain read-modify-write, one writer per element)   */
  TTSK_GEMM_LRELU_OUT = 1 << 10, /* v = leaky_relu(v, out_slope) before the store                           */
  TTSK_GEMM_F16       = 1 << 11, /* 16-bit operands (A, B, C, C2, R, G) are IEEE fp16 instead of bf16           */
  TTSK_GEMM_C2_LRELU  = 1 << 12, /* the second output is leaky_relu(v, out_slope) (C keeps v): the next conv's   */
                                 /* activation is produced by this conv's epilogue instead of its operand staging */
  TTSK_GEMM_DEFER_REDUCE = 1 << 13,/* split-K: write the partial slabs only; the caller sums many GEMMs' slabs later */
                                 /* with ONE ttsk_gemm_reduce_batch launch (weight gradients: needed only by Adam)  */
  TTSK_GEMM_RAW_SLABS = 1 << 14    /* the fp32 partial tiles [splits][nz][M][N] go to `workspace` for ANY splits >= 1, no  */
                                 /* epilogue and no reducer launch: the consumer sums them (ttsk_layernorm_bwd_slabs   */
                                 /* adds the slabs and the residual while it reads its rows); C is not written         */
};

typedef struct ttsk_gemm_desc {
  const void* A;      /* bf16 */
  const void* B;      /* bf16 */
  void* C;            /* bf16 or fp32 */
  void* C2;           /* optional second output, bf16, same ldc/strides as C (NULL = none) */
  const float* bias;  /* [N] or NULL */
  const void* R;      /* residual, layout of C with ldr; bf16 or fp32 */
  const void* G;      /* ReLU gate, bf16, layout of C with ldg */
  int32_t M, N, K;    /* output M x N, K = contraction length per tap */
  int32_t lda, ldb, ldc, ldr, ldg;
  int32_t flags;
  float alpha;
  float in_slope, out_slope;
  /* batch: z = z1 * nz2 + z2 */
  int32_t nz1, nz2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  /* conv-A mode (taps > 0; A not transposed): rows of A and C are (segment, position) pairs, row = s*seg_len + t.
   * Tap j reads A row (t + tap_shift0 + j*tap_dshift) of the same segment, zero outside [0, seg_len);
   * B element k of tap j sits at column j*b_tap_stride + k (B_TR: row offset j*b_tap_stride rows... see gemm.hip) */
  int32_t taps, seg_len, tap_shift0, tap_dshift;
  int64_t b_tap_stride;
  /* B_TR row shift per batch index z2 (conv dW): B row (t + bshift0 + z2*bdshift) within a segment of bseg_len rows */
  int32_t bseg_len, bshift0, bdshift;
  /* output row remap (polyphase ConvTranspose1d): C row for A row (s, t) is s*out_seg + t*out_mul + out_add,
   * skipped when outside [0, out_seg).  out_mul == 0 means identity. */
  int32_t out_seg, out_mul, out_add;
  int32_t out_add_dz;   /* out_add += z2 * out_add_dz: several polyphase phases batched into one launch */
  /* split-K: `splits` > 1 cuts the K chunks into `splits` ranges whose fp32 partial tiles go to `workspace`
   * ([splits][nz][M][N] floats); a second kernel sums them in fixed order and applies the epilogue (deterministic). */
  int32_t splits;      /* 0 = let the library choose (ttsk_gemm_plan) */
  void* workspace;
  int64_t workspace_bytes;
  /* tile configuration: 0 = let the library choose, 1 = 128x128x64 (4 waves, register staging; required by LRELU_IN),
   * 2 = 256x128x64 (8 waves, 3-stage LDS-DMA ring), 3 = 64x128x64 (4 waves, register staging, A untransposed: outputs
   * with few columns get enough workgroups to fill the chip without split-K) */
  int32_t kernel;
  /* batched problems with one bias vector per z1 (the three VariancePredictors of a training step as one launch):
   * bias of batch z1 = bias + z1 * s_bias1 (floats); 0 = one bias for all */
  int64_t s_bias1;
} ttsk_gemm_desc;

int ttsk_gemm(const ttsk_gemm_desc* d, void* stream);
/* Deferred split-K reduction.  A GEMM launched with TTSK_GEMM_DEFER_REDUCE (fp32 C, no epilogue beyond alpha / ACCUM_C,
 * nz1 == 1) leaves `splits` slabs [split][nz2][M][N] in its workspace; ttsk_gemm_reduce_batch sums the slabs of `n` such
 * GEMMs in fixed order (deterministic) into their C (C[z2*sC2 + m*ldc + n] (+)= alpha * sum) with one kernel per 64 items. */
typedef struct ttsk_reduce_item {
  const float* ws;
  float* C;
  int32_t M, N, ldc, nz, splits, accumulate;
  int64_t sC2;
  float alpha;
} ttsk_reduce_item;
int ttsk_gemm_reduce_batch(const ttsk_reduce_item* items, int n, void* stream);

/* Grouped launch of n problems with the same operand layout (A_TR / B_TR / F16 flags) and the same tile configuration
 * (desc.kernel: 1 = 128x128, also the default; 2 = 256x128) as ONE
 * grid — for the many small contractions nothing waits for individually (the weight-gradient GEMMs of a backward pass:
 * 32-256 workgroups each).  group_build validates and plans every descriptor (split-K workspaces as for ttsk_gemm) and
 * writes a table of ttsk_gemm_group_table_bytes(n) bytes into HOST memory (pageable is fine); group_launch copies it into
 * the caller's 16-byte aligned device buffer of the same size through kernel arguments (hipGraph-capturable, no pinned
 * staging), runs the grid and then the reducers of split problems that do not carry TTSK_GEMM_DEFER_REDUCE.  Results are
 * identical to n ttsk_gemm calls with the same kernel / splits. */
int64_t ttsk_gemm_group_table_bytes(int n);
int ttsk_gemm_group_build(const ttsk_gemm_desc* descs, int n, void* host_table, int32_t* total_wgs);
int ttsk_gemm_group_launch(const void* host_table, void* dev_table, void* stream);
/* The same with the grid capped at max_wgs workgroups (0 = no cap; honoured by the 256x128 configuration, kernel = 2): each
 * workgroup walks the table's tiles in steps of the grid size.  A grid of fewer workgroups than CUs (that configuration holds one
 * per CU) leaves the remaining CUs to kernels of a concurrent stream: the FS2 backward runs its weight-gradient group beside the
 * encoder-side dX chain this way. */
int ttsk_gemm_group_launch_capped(const void* host_table, void* dev_table, int max_wgs, void* stream);
/* ttsk_gemm_group_launch_capped in two parts: the table upload (kernel-argument launches; a no-op for an inline table) and the
 * grouped launch that expects the table in dev_table.  The upload may go on another stream, ordered before the launch by the caller,
 * so that it does not sit between two grouped launches on the same stream. */
int ttsk_gemm_group_upload(const void* host_table, void* dev_table, void* stream);
int ttsk_gemm_group_launch_uploaded(const void* host_table, void* dev_table, int max_wgs, void* stream);

/* The FFT block's first position-wise conv, forward: out = [relu](Conv1d(256 -> Cout, k)(x) + bias), 'same' zero padding per
 * utterance.  reference: fs_two/transformer/SubLayers.py:93-101 (w_1 + relu).  x [B*S][256] bf16 (row = utterance*S + frame), w the
 * tap-major bf16 weight (Cout, k, 256) as ttsk_gemm's conv takes it, out [B*S][Cout] bf16.  A window kernel (activation window of
 * 112 frames in LDS, weights L2 -> registers, no barrier in the tap loop) for Cin = 256, Cout % 256 == 0, odd k <= 9; same result
 * as the implicit-GEMM conv up to the order of the fp32 accumulation. */
int ttsk_ffn_conv_supported(int Cin, int Cout, int K);
int ttsk_ffn_conv_fwd(const void* x_bf16, const void* w_bf16, const float* bias, void* out_bf16, int B, int S, int Cin, int Cout, int K,
                      int relu, int packed, void* stream);
/* packed = 1: w is the fragment-major repack [k][8][Cout/16][64][8] written by ttsk_ffn_pack_weight from the tap-major weight (a
 * wave's fragment is then 1 KiB contiguous instead of 16 rows x 64 B). */
int ttsk_ffn_pack_weight(const void* w_bf16, void* packed_bf16, int Cout, int K, void* stream);
int ttsk_ffn_pack_weight_batch(const void* const* w_bf16, void* const* packed_bf16, int n /* <= 16 */, int Cout, int K, void* stream);
/* The same window kernel for Cin = 256 or 512 (the PostNet's Conv1d(512 -> 512, k = 5), fs_two/transformer/Layers.py:85-129): out =
 * [relu](conv(x) + bias), bias may be NULL, out bf16 or (out_f32 = 1, Cin = 512 only: the PostNet keeps its conv outputs in fp32 for
 * BatchNorm) fp32.  The weights are always the fragment-major pack written by ttsk_win_conv_pack_batch from the tap-major storage
 * (Cs, K, Ds): transpose = 0 packs the conv's own weights (Cout = Cs, Cin = Ds); transpose = 1 packs the weights of the conv's INPUT
 * GRADIENT seen as a forward conv on dy with flipped taps (Cout = Ds, Cin = Cs), so that ttsk_win_conv(dy, that pack) = dx. */
typedef struct ttsk_pack_item {
  const void* src;     /* tap-major bf16 weight (Cs, K, Ds) */
  void* dst;           /* fragment-major pack, Cs*K*Ds elements */
  int32_t Cs, K, Ds, transpose;
} ttsk_pack_item;
int ttsk_win_conv_supported(int Cin, int Cout, int K);
/* up to 48 packs of any shapes from a HOST item list (handed over through the kernel's arguments: capturable), one launch */
int ttsk_win_conv_pack_items(const ttsk_pack_item* items, int n, void* stream);
/* the same from an item table that already lives in DEVICE memory (8-byte aligned), any n: what a model with fixed weight and pack
 * addresses calls after every optimizer step (one launch, no upload) */
int ttsk_win_conv_pack_table(const ttsk_pack_item* dev_items, int n, void* stream);
int ttsk_win_conv_pack_batch(const void* const* w_bf16, void* const* packed_bf16, int n /* <= 16 */, int Cs, int K, int Ds, int transpose,
                             void* stream);
/* An input gradient whose contraction is wide (w_1: 1024 channels x 9 taps; q|k|v: 768), as nsplit window convs over 256-channel slices
 * of x [B*S][nsplit*256] in ONE launch: w_packed is the whole transposed pack (Cout, K, nsplit*256), slice sp takes its k-steps of every
 * tap and writes the fp32 slab slabs + sp*B*S*Cout — the raw split-K slabs that ttsk_layernorm_bwd_slabs sums (splits = nsplit,
 * stride = B*S*Cout). */
int ttsk_win_conv_split(const void* x_bf16, const void* w_packed, float* slabs, int nsplit, int B, int S, int Cin_total, int Cout, int K,
                        void* stream);
/* gate_bf16 [B*S][Cout] (may be NULL; bf16 output only): out = gate > 0 ? out : 0 — the ReLU backward of SubLayers.py:96 folded into
 * the w_2 input-gradient conv. */
/* delta_out [B*(Cout/128)][S] with delta_o32 [B*S][Cout] fp32 (both may be NULL; bf16 output, Cout = heads*128): the attention
 * backward's delta = rowsum over a head's 128 columns of out (as stored) * delta_o32, written while the rows are stored — out is then
 * the dO of ttsk_flash_attention_bwd(delta_ready = 1). */
int ttsk_win_conv(const void* x_bf16, const void* w_packed, const float* bias /* may be NULL */, const void* gate_bf16 /* may be NULL */,
                  const float* delta_o32 /* may be NULL */, float* delta_out /* may be NULL */, void* out, int out_f32, int B, int S, int Cin,
                  int Cout, int K, int relu, void* stream);
/* ttsk_win_conv (Cin = 512, fp32 output, no gate / delta) whose output's BatchNorm statistics partials come out of the same kernel:
 * stats[ttsk_win_conv_stats_rows(B, S)][2*Cout] = per-channel sum | sum of squares per (utterance, 64-frame tile) over the rows that
 * exist (t < S, t < frame_limit[0] when given) — the `partials` of ttsk_bn_train_apply, so the PostNet's 512 -> 512 layers need no
 * ttsk_bn_stats_slab launch (reference: Layers.py:133-143, Conv1d -> BatchNorm1d). */
int ttsk_win_conv_stats_rows(int B, int S);
/* Round 5: the PostNet's ends on the same kernel (reference: Layers.py:85-129, convolutions 0 and 4 = Conv1d(80 -> 512, k 5) and
 * Conv1d(512 -> 80, k 5); their input gradients are the mirrored shapes on transposed packs).  Cin = 80: rows of 80 channels, the
 * window and the pack zero-padded to three 32-channel k-steps (ttsk_win_conv_pack_* pads such a pack itself); Cout = 80: five waves
 * of 16 channels.  ttsk_win_conv / _stats / _bnb accept both.  ttsk_win_conv_resid: bf16 output = conv + resid_f32 [B*S][Cout], the
 * fp32 sum of the accumulators and the residual rounded ONCE (round 6: the residual joins the accumulators before the staging tile is
 * packed; round 5 added it to the tile's bf16 values and rounded twice) — conv 0's input gradient + the mel terms' own gradient (fastspeech2.py:104, postnet(output) + output). */
int ttsk_win_conv_resid(const void* x_bf16, const void* w_packed, const float* resid_f32, void* out_bf16, int B, int S, int Cin, int Cout,
                        int K, void* stream);
/* ttsk_win_conv with fp32 output and a bf16 copy of the same rows (round 5: mel_linear, fastspeech2.py:102, Linear(256 -> 80) as a k = 1
 * conv on the five-wave instance: the fp32 mel is what the loss reads and what the PostNet's output is added to, the copy is the PostNet's
 * input).  Shapes as ttsk_win_conv plus (Cin 256, Cout 80). */
int ttsk_win_conv_dual(const void* x_bf16, const void* w_packed, const float* bias, float* out_f32, void* out_bf16, int B, int S, int Cin,
                       int Cout, int K, void* stream);
int ttsk_win_conv_stats(const void* x_bf16, const void* w_packed, const float* bias, float* out_f32, float* stats,
                        const int32_t* frame_limit, int B, int S, int Cin, int Cout, int K, void* stream);
/* ttsk_win_conv (Cin = 512, bf16 output: the input gradient of a PostNet 512 -> 512 conv, run on the transposed pack) that also emits
 * the BatchNorm-BACKWARD statistics partials of the layer below — the layer whose upstream gradient `out` is (Layers.py:133-143
 * backwards: conv_i's input gradient is dL/d(dropout(tanh(BN_{i-1}(yc_{i-1}))))): stats[ttsk_win_conv_stats_rows(B, S)][2*Cout] =
 * per-channel sum of dy | sum of dy * xhat per (utterance, 64-frame tile) over the rows that exist, dy = out * keep / (1-p) * (1 -
 * tanh^2(gamma * xhat + beta)) (use_tanh), xhat = (bn_x - mean) * rstd — the `partials` of ttsk_bn_bwd_apply_slab, so that layer needs no
 * ttsk_bn_bwd_stats_slab launch.  keep: ttsk_bn_train_apply's keep bits (required when p > 0). */
int ttsk_win_conv_bnb(const void* x_bf16, const void* w_packed, void* out_bf16, float* stats, const float* bn_x_f32, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, const uint8_t* keep, float p, int use_tanh,
                      const int32_t* frame_limit, int B, int S, int Cin, int Cout, int K, void* stream);
/* HiFi-GAN's stride-8 upsamplers, ConvTranspose1d(Cin -> Cout, k = 16, stride 8, padding 4) (hifi/models.py:166-176,189 with
 * upsample_rates[i] = 8), on the window-conv kernel: x16 (B, T, Cin) fp16 -> out16 (B, 8T, Cout) fp16.  Output frame 8t + r reads
 * x[t] (weight tap r + 4) and x[t - 1] (r < 4: tap r + 12) or x[t + 1] (r >= 4: tap r - 4): a conv with two pseudo-taps and 8 * Cout
 * phase-major output channels whose rows are the output tensor.  w_packed: ttsk_win_conv_pack_items of the (8 * Cout, 2, Cin) tap-major
 * pseudo-weight W2[r * Cout + co][slot][ci]; bias8: the bias repeated 8 times.  Replaces the polyphase implicit GEMMs of ttsk_gemm. */
int ttsk_hifi_upsample8_supported(int Cin, int Cout);
int ttsk_hifi_upsample8(const void* x16, const void* w_packed, const float* bias8, void* out16, int f16, int B, int T, int Cin, int Cout,
                        void* stream);
/* The same for any ConvTranspose1d(kernel 2 * stride, padding stride / 2) the kernel is instantiated for: stride 8 as above, stride 2
 * (k = 4, padding 1) for 128 -> 64 channels (hifi/models.py: ups[2]).  Output frame stride * t + r reads x[t] (tap r + stride / 2) and
 * x[t - 1] (r < stride / 2: tap r + 3 * stride / 2) or x[t + 1] (tap r - stride / 2); pseudo-weight (stride * Cout, 2, Cin). */
int ttsk_hifi_upsample_win_supported(int Cin, int Cout, int stride);
int ttsk_hifi_upsample_win(const void* x16, const void* w_packed, const float* bias_rep, void* out16, int f16, int B, int T, int Cin, int Cout,
                           int stride, void* stream);
/* HiFi-GAN's conv_pre + the LeakyReLU its only reader applies (hifi/models.py:152,186,188), Conv1d(80 -> Cout, k) on fp16 rows, on the
 * window-conv kernel (96-channel instance, contraction zero-padded): out16 (B, T, Cout) = lrelu(conv(x16 (B, T, 80)) + bias, slope).
 * w_packed: ttsk_win_conv_pack_items of the (Cout, k, 80) tap-major weight.  Replaces ttsk_conv1d's implicit GEMM for this layer. */
int ttsk_hifi_conv_pre_win_supported(int Cin, int Cout, int K);
int ttsk_hifi_conv_pre_win(const void* x16, const void* w_packed, const float* bias, void* out16, int f16, int B, int T, int Cin, int Cout, int K,
                           float slope, void* stream);
/* The same operator, operands and pack (ttsk_hifi_upsample_win's) for Cin = 256, stride 8 on a kernel that loads a 96-frame window once and
 * loops over the 8 * Cout / 256 channel groups inside the workgroup, each group's stores in flight under the next group's MFMAs
 * (hifi/models.py: ups[1], 256 -> 128: one round of 256 workgroups instead of two rounds of 448 window loads).  The bias is the
 * accumulators' initial value here: results equal ttsk_hifi_upsample_win's to the last fp32 bit before the fp16 rounding. */
int ttsk_hifi_upsample_loop_supported(int Cin, int Cout, int stride);
int ttsk_hifi_upsample_loop(const void* x16, const void* w_packed, const float* bias_rep, void* out16, int f16, int B, int T, int Cin, int Cout,
                            int stride, void* stream);

/* Fused sub-layer tail of an FFT block (reference: fs_two/transformer/SubLayers.py:62-63 and :96-99 + Layers.py:29,32):
 *   out = zero_PAD_rows( LayerNorm( dropout_{p_pre, site_pre}( A[M,K] @ W[D,K]^T + bias ) + res ) ),  D = 256 only.
 * A, W, res, out, z_save bf16; z_save (may be NULL) receives the LayerNorm input, mean / rstd [M] its statistics (what
 * ttsk_layernorm_bwd needs); lens (may be NULL) marks rows t >= lens[row / seg_len] of each segment as PAD.  Same dropout
 * bits as ttsk_layernorm_fwd at the same (rng, site) — the two forms are interchangeable up to the bf16 rounding of
 * the GEMM output that the fused kernel skips. */
int ttsk_gemm_ln_fwd(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res, const float* gamma,
                     const float* beta, void* out, void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len,
                     int M, int K, int D, float eps, float p_pre, uint32_t site_pre, const void* rng, void* stream);
/* The same fused op on the window-conv data path: W_packed is the MFMA-fragment-major pack of the (256, K) weight
 * (ttsk_win_conv_pack_* with (Cs, K, Ds) = (256, 1, K), transpose = 0); K = 256 or 1024.  Same row code, same dropout mask. */
int ttsk_win_ln_supported(int K, int D);
int ttsk_win_ln_fwd(const void* A, int lda, const void* W_packed, const float* bias, const void* res, const float* gamma, const float* beta,
                    void* out, void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len, int M, int K, int D, float eps,
                    float p_pre, uint32_t site_pre, const void* rng, void* stream);
/* ttsk_win_ln_fwd whose output rows also go, inside the same kernel, through the NEXT FFTBlock's q|k|v projection
 * (reference SubLayers.py:41-43: w_qs / w_ks / w_vs on the block input): proj_out[M][768] = out · W' + proj_bias, proj_w_packed =
 * the ttsk_win_conv pack of the (768, 1, 256) weight.  One launch instead of two dependent ones; proj_out is bit-identical to
 * ttsk_win_conv on `out`. */
int ttsk_win_ln_proj_fwd(const void* A, int lda, const void* W_packed, const float* bias, const void* res, const float* gamma,
                         const float* beta, void* out, void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len, int M,
                         int K, int D, float eps, float p_pre, uint32_t site_pre, const void* rng, const void* proj_w_packed,
                         const float* proj_bias, int proj_Cout, void* proj_out, void* stream);

/* what ttsk_gemm will run for `d` (with d->kernel / d->splits as constraints when non-zero) and the workspace it needs */
int ttsk_gemm_plan(const ttsk_gemm_desc* d, int32_t* kernel, int32_t* splits, int64_t* workspace_bytes);


/* ------------------------------------------------------------------------------------ LengthRegulator
 * reference: fs_two/model/modules.py:220-252 (LengthRegulator.LR/expand) + fs_two/utils/tools.py:369-387 (pad)
 *            + fs_two/transformer/Models.py:176-178 (decoder adds the position table right after).
 * Integer-exact: di = max(trunc(d), 0); cs = inclusive wavefront prefix scan of di (64 phonemes per scan
 * step); frame t of utterance b copies phoneme #{i : cs[i] <= t} when t < cs[L-1], else it is a zero row;
 * mel_len[b] = cs[L-1] (NOT cropped to T).  `pe` (fp32 [>=T][D], may be NULL) is added to every row.
 * dur_dtype: 0 = int64, 1 = fp32, 2 = int32.
 */
int ttsk_length_regulator_fwd(const void* x_bf16, const void* dur, int dur_dtype, const float* pe, void* out_bf16,
                              int32_t* idx_out /* [B][T], -1 = zero row, may be NULL */,
                              int32_t* cumsum_out /* [B][L] */, int64_t* mel_len /* [B] */, int B, int L, int T, int D,
                              void* stream);
/* dx[b][i][:] = sum over the frames that copied phoneme i of dout[b][t][:]  (segment sum, no atomics) */
int ttsk_length_regulator_bwd(const void* dout_bf16, const int32_t* cumsum, void* dx_bf16, int B, int L, int T, int D,
                              void* stream);

/* ------------------------------------------------------------------------------ LayerNorm (fused block tail)
 * out = mask( dropout_post( LN( dropout_pre(y) + res ) ) ),  optional head_out[row] = mask(<out, head_w> + head_b)
 * reference: SubLayers.py:62-63,99-101 (LN(dropout(sublayer)+residual)), Layers.py:29,32 (masked_fill of PAD rows),
 *            model/modules.py:270-309 (VariancePredictor: ReLU -> LN -> Dropout; Linear(256,1) -> masked_fill).
 * rows are (segment, position) pairs; a row is PAD when position >= lens[segment] (lens may be NULL).
 * `rng` points at {uint64 seed, uint64 step} in device memory; masks are functions of (seed, step, site, element).
 * z_save receives the LN input (bf16) for the backward; mean/rstd are per-row fp32.
 */
int ttsk_layernorm_fwd(const void* y_bf16, const void* res_bf16, const float* gamma, const float* beta, void* out_bf16,
                       void* z_save_bf16, float* mean, float* rstd, const int64_t* lens, int seg_len, int rows, int D,
                       float eps, float p_pre, uint32_t site_pre, float p_post, uint32_t site_post, const uint64_t* rng,
                       const float* head_w, const float* head_b, float* head_out, void* stream);
int ttsk_layernorm_bwd_nblocks(int rows);
/* partials: [nblocks][3*D] = dbias(sum of dy) | dgamma | dbeta, or [nblocks][4*D + 1] with the head (| dhead_w | dhead_b)
 * — the order in which these parameters sit in the flat gradient buffer, so one ttsk_colsum_finalize call lands them all.
 * dz = grad wrt the LN input (bf16; times (z > 0) when relu_in); dy = dz through the pre-dropout mask (only if p_pre > 0). */
int ttsk_layernorm_bwd(const void* dout_bf16, const float* dhead, const float* head_w, const void* z_bf16, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, const int64_t* lens, int seg_len, int rows,
                       int D, int relu_in, float p_pre, uint32_t site_pre, float p_post, uint32_t site_post,
                       const uint64_t* rng, void* dz_bf16, void* dy_bf16, float* partials, void* stream);
/* Grouped forms: rows = groups * group_rows; group g reads gamma / beta / head_w / head_b at + g * param_stride floats and
 * uses dropout sites + g * site_stride; PAD masks (lens [group_rows / seg_len], shared by the groups) and dropout element
 * indices are taken inside the group, so one grouped launch equals `groups` single launches bit for bit (the three
 * VariancePredictors of a training step: model/modules.py:158-193 with targets given are independent).  bwd partials:
 * [groups][ttsk_layernorm_bwd_nblocks(group_rows)][3*D or 4*D+1]. */
int ttsk_layernorm_fwd_grouped(const void* y_bf16, const void* res_bf16, const float* gamma, const float* beta, void* out_bf16,
                               void* z_save_bf16, float* mean, float* rstd, const int64_t* lens, int seg_len, int groups,
                               int group_rows, int64_t param_stride, uint32_t site_stride, int D, float eps, float p_pre,
                               uint32_t site_pre, float p_post, uint32_t site_post, const uint64_t* rng, const float* head_w,
                               const float* head_b, float* head_out, void* stream);
int ttsk_layernorm_bwd_grouped(const void* dout_bf16, const float* dhead, const float* head_w, const void* z_bf16,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, const int64_t* lens,
                               int seg_len, int groups, int group_rows, int64_t param_stride, uint32_t site_stride, int D,
                               int relu_in, float p_pre, uint32_t site_pre, float p_post, uint32_t site_post, const uint64_t* rng,
                               void* dz_bf16, void* dy_bf16, float* partials, void* stream);
/* ttsk_layernorm_bwd whose upstream gradient is still in split-K form: dout[row] = sum_s slabs[s*slab_stride + row*D ..] (+ R[row],
 * bf16, may be NULL) — the raw partial tiles a ttsk_gemm with TTSK_GEMM_RAW_SLABS left behind (the dX GEMM that feeds this
 * LayerNorm) are summed in fixed order while the rows are read, in fp32, so neither a reducer launch nor a bf16 copy of dout
 * exists.  Other arguments as ttsk_layernorm_bwd (no head mode). */
int ttsk_layernorm_bwd_slabs(const float* slabs, int nsplit, int64_t slab_stride, const void* R_bf16, const void* z_bf16,
                             const float* mean, const float* rstd, const float* gamma, const float* beta, const int64_t* lens,
                             int seg_len, int rows, int D, int relu_in, float p_pre, uint32_t site_pre, float p_post,
                             uint32_t site_post, const uint64_t* rng, void* dz_bf16, void* dy_bf16, float* partials, void* stream);
/* ttsk_layernorm_bwd (D = 256, no head / ReLU input / post dropout; upstream gradient as dout_bf16 OR as split-K slabs + R as in
 * ttsk_layernorm_bwd_slabs) followed, in the same kernel, by the k = 1 projection that consumes dy — the sub-layer's input gradient:
 *   out[rows][Cout] = dy · W'   with w_packed = ttsk_win_conv's pack of the transposed weight (Cout = 256 or 1024, Cin = 256),
 *   gate_bf16 (may be NULL): out = gate > 0 ? out : 0  (w_2's dX through the ReLU: reference SubLayers.py:93-101 backward),
 *   delta_o32 / delta_out (may be NULL; Cout = 256 = 2 heads x 128, rows = B*seg_len): the attention backward's
 *   delta[(b*2 + h)*seg_len + t] = sum over head h's columns of out * o32  (fc's dX: SubLayers.py:62-63 backward).
 * pre_x_bf16 / pre_w_packed / pre_K (instead of dout_bf16 and slabs; pre_K = 768): the upstream gradient is itself a k = 1 projection
 * that only this LayerNorm reads, dout = pre_x · W_pre' (+ R) — the input gradient of the FOLLOWING block's q|k|v projection
 * (pre_x = dqkv [rows][768], pre_w_packed = the pack of the transposed (768, 1, 256) weight): computed per 32-row tile in fp32,
 * never written to memory (three dependent launches become one).
 * One launch instead of two dependent ones; results are bit-identical to ttsk_layernorm_bwd(_slabs) + ttsk_win_conv.
 * partials: [ttsk_layernorm_bwd_proj_nblocks(rows)][3*D] (dbias | dgamma | dbeta), a workgroup per 32 rows. */
int ttsk_layernorm_bwd_proj_nblocks(int rows);
int ttsk_layernorm_bwd_proj(const void* dout_bf16, const float* slabs, int nsplit, int64_t slab_stride, const void* R_bf16,
                            const void* z_bf16, const float* mean, const float* rstd, const float* gamma, const int64_t* lens,
                            int seg_len, int rows, int D, float p_pre, uint32_t site_pre, const uint64_t* rng, void* dz_bf16,
                            void* dy_bf16, float* partials, const void* w_packed, int Cout, const void* gate_bf16,
                            const float* delta_o32, float* delta_out, void* out_bf16, const void* pre_x_bf16,
                            const void* pre_w_packed, int pre_K, void* stream);
/* The q|k|v input gradient on its own, out[rows][256] = dqkv[rows][768] · W' (+ R_bf16, may be NULL), for the first block of a stack
 * (no LayerNorm backward in front of it to host the product: reference SubLayers.py:41-43 backward + the residual of :62): 32-row
 * tiles, one pass over the contraction, the window-conv pack of the transposed weight, no split-K slabs.  K = 768, D = 256. */
int ttsk_qkv_dx(const void* dqkv_bf16, const void* w_packed, const void* R_bf16, void* out_bf16, int rows, int K, int D, void* stream);
/* dst[c] (+)= scale * sum_b partials[b*ld + c]  in fixed order */
int ttsk_colsum_finalize(const float* partials, int nblk, int ncols, int ld, float* dst, int accumulate, float scale,
                         void* stream);
/* the same for a list of (partials, dst) pairs in one launch per 64 items (gradient column sums are needed only by Adam) */
typedef struct ttsk_finalize_item {
  const float* partials;
  float* dst;
  int32_t nblk, ncols, ld, accumulate;
  float scale;
} ttsk_finalize_item;
int ttsk_colsum_finalize_batch(const ttsk_finalize_item* items, int n, void* stream);
int ttsk_colsum_nblocks(int rows);
/* per-block column sums of x [rows][C] (bf16, or fp32 when is_f32) -> partials[nblocks][C]  (bias gradients) */
int ttsk_colsum(const void* x, int is_f32, int rows, int C, int ld, float* partials, void* stream);
/* the same for up to 64 matrices per launch; nblk must be ttsk_colsum_nblocks(rows) */
typedef struct ttsk_colsum_item {
  const void* x;
  float* partials;
  int32_t is_f32, rows, C, ld, nblk;
} ttsk_colsum_item;
int ttsk_colsum_batch(const ttsk_colsum_item* items, int n, void* stream);

/* Training-mode VarianceAdaptor embedding chain in one pass (reference: model/modules.py:158-193 with targets given;
 * fastspeech2.py:72-75): x1 = x + speaker_table[speakers[row / L]]; x2 = x1 + pitch_table[bucketize(pitch_target)];
 * x3 = x2 + energy_table[bucketize(energy_target)] (bucketize = torch.bucketize, right=False, n_bins_minus_1 edges), each
 * rounded to bf16; the bucket indices are returned for the embedding-gradient scatter-sums.  va_combine is its backward around
 * the grouped predictor backward: dxin [3][rows][D] fp32 (gradients wrt x, x1, x2 from the duration / pitch / energy
 * predictors), dx3 -> dx2 = dx3 + dxin[2], dx1 = dx2 + dxin[1], dx = dx1 + dxin[0].
 * row_limit (int64 [rows / L], may be NULL): phoneme positions l >= row_limit[u] of utterance u do not exist in the reference's
 * batch (shape-bucketed training pads L up to a multiple of 8): they are written as zero rows / carry no gradient, so that the
 * predictors' convolutions see the zero padding the reference's shorter batch has there. */
int ttsk_va_embed(const void* x_bf16, const float* speaker_table, const int64_t* speakers, int L, const float* pitch_target,
                  const float* pitch_bins, const float* pitch_table, const float* energy_target, const float* energy_bins,
                  const float* energy_table, int n_bins_minus_1, void* x1_bf16, void* x2_bf16, void* x3_bf16, int32_t* pitch_idx,
                  int32_t* energy_idx, int rows, int D, const int64_t* row_limit, void* stream);
int ttsk_va_combine(const void* dx3_bf16, const float* dxin, void* dx2_bf16, void* dx1_bf16, void* dx_bf16, int rows, int D, int L,
                    const int64_t* row_limit, void* stream);

/* ------------------------------------------------------------------------------------------ attention softmax
 * reference: fs_two/transformer/Modules.py:15-22.  scores fp32 [nz][S][Sp] (already scaled by 1/sqrt(d_k) in the
 * Q.K^T GEMM epilogue), z = b*H + h; keys >= lens[b] get -inf; probs bf16 [nz][S][Sp] with zero pad columns.
 * bwd: dscores = alpha * P o (dP - rowsum(dP o P)).
 */
int ttsk_softmax_fwd(const float* scores, void* probs_bf16, const int64_t* lens, int nz, int H, int S, int Sp, void* stream);
int ttsk_softmax_bwd(const void* probs_bf16, const float* dprobs, void* dscores_bf16, int nz, int S, int Sp, float alpha,
                     void* stream);

/* ---------------------------------------------------------------------------- flash attention (d_k = 128)
 * softmax(q k^T * scale, keys >= lens[b] masked) v per (utterance, head) on the fused projection output qkv [B*S][3*d] (q | k | v, head h =
 * columns h*128.. of each part), heads merged back in o [B*S][d] (reference: fs_two/transformer/Modules.py:14-24 + SubLayers.py:44-60)
 * without any S x S tensor in HBM: the forward keeps a running row max / sum over 64-key tiles and returns O and, for the
 * backward, lse [B*H][S] = log sum_k exp(score); the backward recomputes P = exp(score - lse) per tile.
 * bwd: delta_ws [B*H][S] fp32 scratch; writes ALL of dqkv [B*S][3*d] (dQ | dK | dV, head h at columns h*128 of each part),
 * two launches (delta = rowsum(dO o O); then the query side (dQ) and the key side (dK, dV) as one grid), no atomics.
 * o_f32 [B*S][d] (may be NULL): O before its rounding to bf16.  The backward's delta = rowsum(dO o O) is what dP is cancelled
 * against; taken from the bf16 O its 2^-9 error dominates small dQ / dK gradients, so a training forward keeps the fp32 copy. */
int ttsk_flash_attention_fwd(const void* qkv_bf16, void* o_bf16, float* o_f32 /* may be NULL */, float* lse /* may be NULL */,
                             const int64_t* lens, int B, int H, int S, int d, float scale, void* stream);
/* delta_ready = 1: delta_ws already holds delta (the producer of dout wrote it: ttsk_win_conv with delta_out), no delta launch. */
int ttsk_flash_attention_bwd(const void* qkv_bf16, const void* o_bf16, const float* o_f32 /* may be NULL */, const void* dout_bf16,
                             const float* lse, float* delta_ws, int delta_ready, void* dqkv_bf16, const int64_t* lens, int B, int H, int S,
                             int d, float scale, void* stream);

/* ------------------------------------------------------------------------------- embeddings / variance adaptor
 * bucketize: idx = #{bins < v*scale} (torch.bucketize right=False; reference: model/modules.py:95-100,134-139)
 * gather_add: out[row] = (in ? in[row] : 0) + table[idx[row / idx_div]] + (pe ? pe[row % pe_mod] : 0)
 *             reference: Models.py:101-103, fastspeech2.py:72-75 + modules.py:159, modules.py:95-100,134-139
 * scatter_sum: dtable[v] (+)= sum of dx rows whose index is v, ascending row order, no atomics; skip_row = padding_idx
 */
int ttsk_bucketize(const float* values, const float* bins, int n_bins, float scale, int32_t* idx, float* scaled_out,
                   int n, void* stream);
/* d = clamp(round(exp(logd) - 1) * d_control, min 0), fp32 — reference: model/modules.py:199-203 */
int ttsk_duration_round(const float* logd, float d_control, float* out, int n, void* stream);
/* mask[b][t] = (t >= lens[b]) as bytes (torch.bool storage) — reference: fs_two/utils/tools.py:121-131 */
int ttsk_length_mask(const int64_t* lens, uint8_t* mask, int B, int T, void* stream);
/* out = a + scale_b * b */
int ttsk_add_f32(const float* a, const float* b, float scale_b, float* out, int64_t n, void* stream);
int ttsk_gather_add(const void* in_bf16, const float* table, const void* idx, int idx_is_i64, int idx_div, const float* pe,
                    int pe_mod, void* out_bf16, int rows, int D, void* stream);
int ttsk_scatter_sum(const void* dx_bf16, const void* idx, int idx_is_i64, int idx_div, int n_idx, float* dtable,
                     int n_table_rows, int D, int skip_row, int accumulate, void* stream);
/* the same for up to 8 independent tables per launch (all embedding-table gradients of a backward pass) */
typedef struct ttsk_scatter_item {
  const void* dx;      /* (n_idx * idx_div, D) bf16 */
  const void* idx;
  float* dtable;       /* (n_table_rows, D) fp32 */
  int32_t idx_is_i64, idx_div, n_idx, n_table_rows, D, skip_row, accumulate;
} ttsk_scatter_item;
int ttsk_scatter_sum_batch(const ttsk_scatter_item* items, int n, void* stream);

/* ------------------------------------------------------------------------------------------------ conversions */
int ttsk_cast_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);
int ttsk_nct_to_ntc(const float* src, void* dst16, int f16, int B, int C, int T, void* stream);  /* f16: 0 = bf16, 1 = fp16 */
/* (x * scale) truncated toward zero to int16 — reference: hifiapi.py:50-51 */
int ttsk_to_int16(const float* src, int16_t* dst, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------------------------ HiFi-GAN generator
 * reference: hifi/models.py:146-210 (Generator), :12-95 (ResBlock1), hifi/vocoder/utils.py:24-37.
 * 16-bit tensors of this family are bf16 (f16 = 0) or IEEE fp16 (f16 = 1; what hifigan.py uses: the generator is
 * inference-only, fp16 has 3 more mantissa bits than bf16 at the same MFMA rate and its activations stay far inside
 * fp16's range).  The convolutions themselves run on ttsk_gemm (conv-A mode: dilation = tap_dshift, LeakyReLU fused into the
 * operand staging, ConvTranspose1d as `stride` polyphase launches with the output-row remap).
 * weight_norm_fold: w[r][:] = v[r][:] * g[r] / ||v[r][:]||  (remove_weight_norm, dim 0 — for ConvTranspose1d rows are
 *                   IN-channels, hifi/models.py:203-210).
 * pack_conv_weight: mode 0: Conv1d (Cout,Cin,k) fp32 -> (Cout,k,Cin) bf16; mode 1: ConvTranspose1d (Cin,Cout,k) fp32 ->
 *                   (k,Cout,Cin) bf16.  (d0,d1,d2) = the source shape.
 * avg3:             out = leaky_relu((a + b + c) * scale, slope) — the multi-receptive-field average, hifi/models.py:190-196,
 *                   with the consumer's LeakyReLU fused (slope 1.0 = none).
 */
int ttsk_weight_norm_fold(const float* v, const float* g, float* w, int rows, int cols, void* stream);
int ttsk_pack_conv_weight(const float* src, void* dst16, int f16, int d0, int d1, int d2, int mode, void* stream);
/* "Window" Conv1d for the C = 128 stage (one conv per launch, activation window resident in LDS, weights streamed from a
 * ttsk_pack_resblock_weight pack): out = [lrelu](conv_{K,dil}(x) + bias [+ R]); out2 (optional) = lrelu(out, slope).
 * x must already be activated (zeros outside the utterance are the conv padding).  reference: hifi/models.py:88-95. */
int ttsk_hifi_conv_window_supported(int C, int K, int dil);
int ttsk_hifi_conv_window(const void* x16, const void* w_pack, const float* bias, const void* R16, void* out16, void* out2_16,
                          int f16, int B, int len, int C, int K, int dil, int lrelu_out, float slope, void* stream);
/* The (c1 dilated -> LeakyReLU -> c2 -> + x) pair of ResBlock1 (hifi/models.py:88-95) as ONE launch at C = 128: x is the raw block
 * input, out = c2(lrelu(c1(lrelu(x)) + b1)) + b2 + x; lrelu(c1 ..) stays in LDS.  Bit-identical to two ttsk_hifi_conv_window
 * launches, a third of their HBM traffic.  w*_pack as for ttsk_hifi_conv_window; out must not alias x.
 * mode folds the MRF average (hifi/models.py:190-197) into a block's last pair, with ttsk_hifi_resblock1's meaning:
 * 0: out = y   1: out += y   2: out = lrelu((out + y) * scale, final_slope).
 * C = 128: a wave owns 32 output channels; C = 64 / 32 (the last two stages): a wave owns a quarter of the frames. */
int ttsk_hifi_conv_pair_supported(int C, int K, int dil);
int ttsk_hifi_conv_pair(const void* x16, const void* w1_pack, const float* bias1, const void* w2_pack, const float* bias2, void* out16,
                        int f16, int B, int len, int C, int K, int dil, float slope, int mode, float scale, float final_slope,
                        void* stream);
/* The same pair at C = 64 (and at C = 128 for K = 3) with the weights STATIONARY in registers (round 6, csrc/pairws.hip; hifi/models.py:88-95,
 * :190-197): one persistent 8-wave workgroup per CU walks a contiguous run of 192-frame tiles (96 at C = 128); four waves hold c1's weights and four c2's for the whole
 * launch (32 output channels x K taps x 64 input channels = 16 K registers per lane), the c1 waves produce tile s's lrelu(c1) window in LDS
 * while the c2 waves consume tile s - 1's; the next tile's x window is in flight under the MFMAs.  Arguments, modes and results as
 * ttsk_hifi_conv_pair (bit-identical to it); C = 64: K in {3, 7, 11}, C = 128: K = 3; dil in {1, 3, 5}; tensors below 2 GiB (32-bit buffer offsets);
 * max_wgs: grid cap (0 = one workgroup per CU). */
int ttsk_hifi_conv_pair_ws_supported(int C, int K, int dil);
int ttsk_hifi_conv_pair_ws(const void* x16, const void* w1_pack, const float* bias1, const void* w2_pack, const float* bias2, void* out16,
                           int f16, int B, int len, int C, int K, int dil, float slope, int mode, float scale, float final_slope,
                           int max_wgs, void* stream);
/* conv_post + tanh (hifi/models.py:198-199): x (B, len, C) 16-bit (already activated), w (1, k, C) tap-major 16-bit,
 * out (B, 1, len) fp32.  A streaming kernel: one output sample per thread. */
int ttsk_hifi_conv_post(const void* x16, const void* w16, const float* bias, float* out, int f16, int B, int len, int C, int K,
                        void* stream);
/* The stride-2, kernel-4 ConvTranspose1d upsamplers (hifi/models.py:166-176,189) as one streaming kernel: both output
 * phases from one read of the input.  x16 (B, T, Cin) 16-bit, w16 (4, Cout, Cin) tap-major (ttsk_pack_conv_weight mode 1),
 * out16 (B, 2T, Cout).  Instances: Cin -> Cout = 128 -> 64, 64 -> 32. */
int ttsk_hifi_upsample2_supported(int Cin, int Cout, int stride, int k);
int ttsk_hifi_upsample2(const void* x16, const void* w16, const float* bias, void* out16, int f16, int B, int T, int Cin,
                        int Cout, void* stream);
int ttsk_avg3(const void* a, const void* b, const void* c, void* out, int f16, int64_t n, float scale, float slope, void* stream);

/* Fused ResBlock1 (hifi/models.py:88-95): all six convs of one block for C in {32,64}, K in {3,7,11}; x/out 16-bit
 * channels-last (B, len, C); weights/biases in the order convs1[0], convs2[0], convs1[1], convs2[1], convs1[2],
 * convs2[2], each weight a fragment-major pack made by ttsk_pack_resblock_weight from the folded (C, C, K) fp32 tensor.
 * mode 0: out = y; 1: out += y; 2: out = (out + y) * scale — the MRF sum / average over the three blocks of a stage
 * (hifi/models.py:190-196); final_slope != 1 applies LeakyReLU(final_slope) to the stored value (the consumer's
 * activation, hifi/models.py:188,197, fused into the last block of the stage). */
int64_t ttsk_resblock_pack_elems(int C, int K);   /* elements of one pack (K padded to the kernel's weight stage) */
int ttsk_pack_resblock_weight(const float* src, void* dst16, int f16, int C, int K, void* stream);
int ttsk_hifi_resblock1(const void* x16, void* out16, int f16, const void* const* weights, const float* const* biases,
                        const int32_t* dilations, int B, int len, int C, int K, int mode, float scale, float slope,
                        float final_slope, void* stream);
int ttsk_hifi_resblock1_supported(int C, int K);
/* The WHOLE last stage of the generator in one launch (hifi/models.py:190-199, csrc/mrf32.hip): the three ResBlock1s of the C = 32
 * multi-receptive-field fusion (kernel sizes k0, k1, k2 = 3, 7, 11) on the same raw input x16 (B, len, 32), their sum * scale (1/3),
 * LeakyReLU(final_slope = 0.01: F.leaky_relu's default, :197), conv_post (32 -> 1, k_post = 7; w_post16 (1, 7, 32) tap-major 16-bit as
 * for ttsk_hifi_conv_post) and tanh: out (B, 1, len) fp32.  weights / biases: 18 entries, block j's six convs at [6 j ..] in
 * ttsk_hifi_resblock1's order and packs; dilations: 3 x 3.  stage_out16 (optional, (B, len, 32)): the activated average that
 * conv_post reads, for tests.  Bit-identical to three ttsk_hifi_resblock1 launches (modes 0, 1, 2) + ttsk_hifi_conv_post; one read of
 * x and one write of the waveform instead of nine tensor passes. */
int ttsk_hifi_mrf32_post_supported(int C, int k0, int k1, int k2, int k_post);
int ttsk_hifi_mrf32_post(const void* x16, float* out, void* stage_out16, int f16, const void* const* weights, const float* const* biases,
                         const int32_t* dilations, const void* w_post16, const float* b_post, int B, int len, int C, int k0, int k1, int k2,
                         int k_post, float slope, float final_slope, float scale, void* stream);

/* rows (u, t) with t >= frame_limit[0] of x [rows][C] (elem_bytes 2 or 4, row = u*seg_len + t) are set to zero: the mel
 * frames / mel gradients past the batch's own longest utterance under shape-bucketed training (see BatchNorm below). */
int ttsk_zero_frames_from(void* x, int elem_bytes, int rows, int C, int seg_len, const int32_t* frame_limit, void* stream);

/* ------------------------------------------------------------------------------------- PostNet BatchNorm1d
 * reference: fs_two/transformer/Layers.py:133-143 — training statistics over ALL rows (PAD rows included),
 * eps 1e-5, momentum 0.1, running_var updated with the unbiased variance; tanh (all but the last layer) and
 * F.dropout(0.5) follow; the last layer adds the mel residual (fastspeech2.py:104).
 * x (the conv output) may be fp32: channels whose batch std is far below their mean amplify a bf16 rounding of x
 * by mean/std, so the model keeps x in fp32 (x_is_f32 = 1).
 * frame_limit (device int32[1], may be NULL) + seg_len: rows are (utterance, frame) pairs, row = u*seg_len + t; frames
 * t >= frame_limit[0] do not exist in the reference's batch (shape-bucketed training pads T up to a multiple of 32 so that
 * hipGraphs repeat; the reference pads to the batch's own longest utterance): they are left out of the statistics and of
 * the row count, are written as zero rows (the next convolution's zero padding) and carry no gradient.
 */
int ttsk_bn_nblocks(int rows);
int ttsk_bn_stats(const void* x, int x_is_f32, int rows, int C, float* partials /* [nblocks][2C] */, const int32_t* frame_limit,
                  int seg_len, void* stream);
int ttsk_bn_finalize(const float* partials, int nblk, int C, int rows, float eps, float momentum, float* mean, float* rstd,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, const int32_t* frame_limit, int seg_len,
                     void* stream);
int ttsk_rsqrt_eps(const float* var, float eps, float* rstd, int n, void* stream);
/* The training path in two launches per direction instead of three.  A workgroup owns a slab of channels (64, or all C when C is
 * not a multiple of 64; then C <= 128) and one of ttsk_bn_nchunks(rows) <= 64 row chunks, so there are few partial rows whatever
 * the grid; ttsk_bn_train_apply / ttsk_bn_bwd_apply_slab sum them for their own slab (fixed order, double) instead of waiting
 * for ttsk_bn_finalize / a column-sum launch.  ttsk_bn_train_apply = ttsk_bn_finalize (mean, rstd, running statistics,
 * num_batches_tracked: all written) + ttsk_bn_apply; ttsk_bn_bwd_apply_slab = the column sums + ttsk_bn_bwd_apply.
 * partials: [nblk][2C]; any producer of such rows will do (nblk need not be ttsk_bn_nchunks).
 * keep_out / keep (may be NULL): uint8 [rows][C/4], bit e of a byte = channel 4*q + e was kept by the dropout; written by the
 * forward and read by the two backward kernels instead of regenerating the Philox mask twice (NULL: they regenerate it). */
int ttsk_bn_nchunks(int rows);
int ttsk_bn_stats_slab(const void* x, int x_is_f32, int rows, int C, float* partials /* [nchunks][2C] */, const int32_t* frame_limit,
                       int seg_len, void* stream);
int ttsk_bn_train_apply(const void* x, int x_is_f32, const float* partials, int nblk, float eps, float momentum, float* mean,
                        float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                        const float* beta, int rows, int C, int use_tanh, float p, uint32_t site, const uint64_t* rng,
                        const float* resid_f32, void* out_bf16, float* out_f32, uint8_t* keep_out, const int32_t* frame_limit,
                        int seg_len, void* stream);
int ttsk_bn_bwd_stats_slab(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                           const uint64_t* rng, const uint8_t* keep, float* partials /* [nchunks][2C] */,
                           const int32_t* frame_limit, int seg_len, void* stream);
int ttsk_bn_bwd_apply_slab(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                           const uint64_t* rng, const uint8_t* keep, const float* partials, int nblk, void* dx_bf16, float* dgamma,
                           float* dbeta, int accumulate /* 1: dgamma / dbeta += ; 0: = */, const int32_t* frame_limit, int seg_len,
                           void* stream);
int ttsk_bn_apply(const void* x, int x_is_f32, const float* mean, const float* rstd, const float* gamma, const float* beta, int rows,
                  int C, int use_tanh, float p, uint32_t site, const uint64_t* rng, const float* resid_f32, void* out_bf16,
                  float* out_f32, const int32_t* frame_limit, int seg_len, void* stream);
int ttsk_bn_bwd_stats(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                      const uint64_t* rng, float* partials /* [nblocks][2C]: sum dy | sum dy*xhat */, const int32_t* frame_limit,
                      int seg_len, void* stream);
int ttsk_bn_bwd_apply(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                      const uint64_t* rng, const float* sums /* [2C] */, void* dx_bf16, float* dgamma, float* dbeta,
                      int accumulate /* 1: dgamma / dbeta += ; 0: = */, const int32_t* frame_limit, int seg_len, void* stream);

/* ------------------------------------------------------------------------------------------------- loss
 * reference: fs_two/model/loss.py:24-134 (use_cwt False).  losses[8] = {total, mel_total, pitch, energy, duration,
 * 0, 0, n_valid_phonemes}; gradients of grad_scale*total: dmel_sum = d/dmel + d/dpost (postnet adds mel back),
 * dpost, dpitch, denergy, dlogd.  partials: [ttsk_fs2_loss_nblocks()][6] fp32.
 * frame_limit (device int32[1], may be NULL): the mel terms are means over B * frame_limit[0] * n_mel elements instead of
 * B * T * n_mel — the reference's denominator when the batch was padded beyond its own longest utterance (see BatchNorm).
 */
int ttsk_fs2_loss_nblocks(void);
int ttsk_fs2_loss(const float* mel, const float* post, const float* mel_target, const int64_t* mel_lens, const float* pitch,
                  const float* energy, const float* logd, const float* pitch_target, const float* energy_target,
                  const int64_t* dur_target, const int64_t* src_lens, int B, int T, int T_target, int n_mel, int L,
                  float grad_scale, float* dmel_sum, float* dpost, float* dpitch, float* denergy, float* dlogd,
                  float* partials, float* losses, const int32_t* frame_limit, void* stream);
/* The same loss as three launches (a `partials` buffer per half: each stream writes memory of its own; they may be one buffer), for a step whose variance predictors run on a stream of their own: the
 * frame-level terms (loss.py:57-77: mel MSE + L1, postnet L1) and dmel_sum / dpost read nothing of the predictors; the phoneme-level terms
 * (loss.py:79-99) and dpitch / denergy / dlogd nothing of the decoder; ttsk_fs2_loss_finalize turns both sets of partial rows into
 * losses[8] wherever both are visible.  Same grid and per-thread order as ttsk_fs2_loss: bit-identical losses and gradients. */
int ttsk_fs2_loss_mel(const float* mel, const float* post, const float* mel_target, const int64_t* mel_lens, int B, int T, int T_target,
                      int n_mel, float grad_scale, float* dmel_sum, float* dpost, float* partials, const int32_t* frame_limit, void* stream);
int ttsk_fs2_loss_var(const float* pitch, const float* energy, const float* logd, const float* pitch_target, const float* energy_target,
                      const int64_t* dur_target, const int64_t* src_lens, int B, int L, float grad_scale, float* dpitch, float* denergy,
                      float* dlogd, float* partials, void* stream);
int ttsk_fs2_loss_finalize(const float* partials_mel, const float* partials_var, const int64_t* src_lens, int B, int T, int n_mel,
                           const int32_t* frame_limit, float* losses, void* stream);

/* ------------------------------------------------------------------------------------------- optimiser
 * reference: train.py:47-54, fs_two/model/optimizer.py:5-53, torch.optim.Adam.
 * `state` is a device block of ttsk_optim_state_bytes() bytes:
 *   { int64 sched_step; int64 adam_t; uint64 rng_seed; uint64 rng_step; float lr, bc1, bc2, clip_coef, gnorm; ... }
 * (its address + 16 is the `rng` pointer handed to the dropout kernels).
 * optim_advance: ++sched_step, ++adam_t, recompute lr/bc1/bc2 on device (graph-replay safe).
 * clip_adam_step: ||g|| -> clip_coef = min(1, max_norm/(||g||+1e-6)) -> Adam on the flat buffers, writes the bf16
 * weight shadow and (optionally) zeroes the gradients.  n % 4 == 0.  partials: 1024 floats.
 */
int ttsk_optim_state_bytes(void);
int ttsk_optim_advance(void* state, float d_model, float warmup, const float* anneal_steps_host, int n_anneal,
                       float anneal_rate, float beta1, float beta2, void* stream);
int ttsk_rng_advance(void* state, void* stream);
/* The keep-mask of dropout site `site` at the state `rng` points to ({uint64 seed, uint64 step}): keep[e] = 1 iff element e (row-major
 * index into the site's [rows][C] tensor) survives with drop probability p — the very function of (seed, step, site, e) every kernel of
 * the step evaluates (masks are regenerated, never stored).  Sites of the FS2 step: encoder block i: 2 i (fc), 2 i + 1 (w_2); decoder
 * block i: 100 + 2 i, 101 + 2 i; predictor g in (duration, pitch, energy): 200 + 2 g, 201 + 2 g; PostNet layer i: 300 + i
 * (reference sites: SubLayers.py:62,98; modules.py:283,295; Layers.py:137-140).  n % 4 == 0.  For parity tests. */
int ttsk_dropout_keep_mask(const uint64_t* rng, uint32_t site, int64_t n, float p, uint8_t* keep, void* stream);
int ttsk_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n,
                        void* state, float* partials, float max_norm, float beta1, float beta2, float eps, int zero_grad,
                        void* stream);
int ttsk_grad_sumsq(const float* grads, int64_t n, float* partials, void* stream);
/* optim_advance (+ rng_advance when advance_rng) + clip_adam_step as TWO launches: the first sums g^2 per block and advances
 * the counters, the second derives the clip coefficient from the partials in every workgroup and applies Adam.  Same
 * arithmetic, same state block as the three calls it replaces. */
int ttsk_optim_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n, void* state,
                    float* partials, float max_norm, float beta1, float beta2, float eps, int zero_grad, float d_model, float warmup,
                    const float* anneal_steps_host, int n_anneal, float anneal_rate, int advance_rng, void* stream);

/* Weight gradient of a Conv1d with K taps ("same" zero padding, dilation 1) as a kernel of its own (csrc/dwconv.hip):
 *   dw[co][tap][ci] (+)= sum_b sum_{t < len_b} dy[(b*S + t)*ldy + co] * x[(b*S + t + tap - K/2)*ldx + ci],   0 <= t + tap - K/2 < S
 * dy / x bf16 rows of B utterances x S rows, dw fp32 (Cout, K, Cin) — the tap-major storage of the flat gradient buffer; lens (int64
 * [B], may be NULL = S): rows t >= lens[b] of dy are not walked (PAD rows carry no gradient: fs_two/transformer/Layers.py:29,32).
 * One launch for up to 12 problems of the same K.  Replaces the K batched problems per weight that ttsk_gemm_group_* ran for
 * torch's conv weight gradient (reference call sites: SubLayers.py:96 w_1 with K = 9, Layers.py:85-129 PostNet with K = 5). */
typedef struct ttsk_dwconv_item {
  const void* dy;
  const void* x;
  float* dw;
  const int64_t* lens;
  int32_t Cout, Cin, K, ldy, ldx, B, S, accumulate;
} ttsk_dwconv_item;
int ttsk_dwconv_supported(int Cout, int Cin, int K);
int ttsk_dwconv_batch(const ttsk_dwconv_item* items, int n /* <= 12 */, void* stream);

/* The same weight gradient for Cout and Cin multiples of 256 (a Linear's or a k = 1 conv's: K = 1; or a conv with taps) on the
 * 256 x 256-tile kernel of csrc/dwgemm.hip — one workgroup per (output tile, tap, utterance range): up to 28 problems per launch.
 * splits > 1 cuts the utterances into `splits` ranges whose partial sums go to `workspace` (fp32 [splits][K][Cout][Cin],
 * ttsk_dwgemm_workspace_floats) — the slab layout of ttsk_reduce_item {ws, C = dw, M = Cout, N = Cin, ldc = K*Cin, nz = K, splits,
 * accumulate, sC2 = Cin}: run ttsk_gemm_reduce_batch behind it; splits = 1 writes / accumulates dw directly.
 * reference call sites: SubLayers.py:41-43,62 (w_qs | w_ks | w_vs, fc), :97 (w_2), Layers.py:85-129 (PostNet convs). */
typedef struct ttsk_dwgemm_item {
  const void* dy;
  const void* x;
  float* dw;
  float* workspace;
  const int64_t* lens;
  int32_t Cout, Cin, K, ldy, ldx, B, S, accumulate, splits;
} ttsk_dwgemm_item;
int ttsk_dwgemm_supported(int Cout, int Cin, int K);
int64_t ttsk_dwgemm_workspace_floats(int Cout, int Cin, int K, int splits);
/* max_wgs > 0: at most that many workgroups, each walking several tiles (one per CU on part of the chip, the rest left to a concurrent stream) */
int ttsk_dwgemm_batch(const ttsk_dwgemm_item* items, int n /* <= 28 */, int max_wgs, void* stream);

/* ttsk_optim_step whose Adam launch also writes the window kernels' weight packs (no ttsk_win_conv_pack_table launch after the step).
 * dev_items [n_items] (device memory, sorted by tile0): the packed weights — tap-major storage (Cs, K, Ds) at element offset `off` of the
 * flat buffers, Ds % 256 == 0 and Cs % 32 == 0, its plain pack (`pack`, the weight as it is: Cout = Cs, Cin = Ds) and / or its transposed,
 * tap-flipped pack (`pack_t`: Cout = Ds, Cin = Cs), either may be NULL; a tile = 32 storage rows x 256 storage columns of one tap,
 * item i owns tiles [tile0, tile0 + K * (Cs/32) * (Ds/256)).  dev_gaps [n_gaps][3] (device, int64): {start, end, sum of (end - start) / 4
 * over the gaps before this one} for the element ranges outside every item, each start / end a multiple of 4; gap_floats = their total.
 * n_tiles * 8192 + gap_floats must equal n.  Parameters, moments, shadow and packs come out bit-identical to ttsk_optim_step followed by
 * ttsk_win_conv_pack_table.  reference: train.py:47-54, fs_two/model/optimizer.py:35-53, torch.optim.Adam. */
typedef struct ttsk_adam_item {
  int64_t off;
  void* pack;
  void* pack_t;
  int32_t Cs, K, Ds, tile0;
} ttsk_adam_item;
int ttsk_optim_step_packed(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n, void* state,
                           float* partials, float max_norm, float beta1, float beta2, float eps, int zero_grad, float d_model, float warmup,
                           const float* anneal_steps_host, int n_anneal, float anneal_rate, int advance_rng,
                           const ttsk_adam_item* dev_items, int n_items, int n_tiles, const int64_t* dev_gaps, int n_gaps,
                           int64_t gap_floats, void* stream);

/* ------------------------------------------------------------------------------------------- mel extraction
 * SURVEY.md §8 row f-3.  reference: hifi/meldataset.py:49-74 (mel_spectrogram), fs_two/audio/stft.py:57-90
 * (STFT.transform: strided conv with a windowed Fourier basis), :174-193 (TacotronSTFT.mel_spectrogram).
 * The contraction is a ttsk_gemm conv (taps = n_fft/hop over rows of `hop` samples); these are the kernels around it.
 * stft_frames: wav (B, len) fp32 -> hop-block rows out16 fp16 (B, rows, 3*hop) = [hi | hi | lo]: reflect padding by `pad`
 *   samples on both sides, zeros past len + 2*pad, value*scale split as hi + lo (fp16 each); against a basis laid out
 *   [hi | lo | hi] per tap one contraction gives hi*hi + hi*lo + lo*hi.  hop % 8 == 0, pad < len.
 * mel_from_spec: spec (B*rows, ld) fp32 with Re in columns [0, nbins) and Im in [nbins, 2*nbins) ->
 *   mel (B, n_mels, T) = log(max(basis . sqrt(re^2 + im^2 + eps), clip)), energy (B, T) = sqrt(sum re^2 + im^2).
 *   The filterbank is passed packed: filter m = basis_vals[basis_off[m] .. basis_off[m+1]) applied to bins
 *   basis_start[m] ..; nnz = basis_off[n_mels].  nbins <= 1088, n_mels <= 80. */
int ttsk_stft_frames(const float* wav, void* out16, int B, int len, int pad, int rows, int hop, float scale, void* stream);
int ttsk_mel_from_spec(const float* spec, int ld, const float* basis_vals, const int32_t* basis_start,
                       const int32_t* basis_off, int nnz, float* mel, float* energy, int B, int rows, int T, int nbins,
                       int n_mels, float eps, float clip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTSK_H */
