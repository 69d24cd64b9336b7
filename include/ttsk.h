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
  TTSK_GEMM_LRELU_OUT = 1 << 10  /* v = leaky_relu(v, out_slope) before the store                           */
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
  /* split-K: `splits` > 1 writes fp32 partial slabs C + split*sCs (C_F32 required, no epilogue except alpha) */
  int32_t splits;
  int64_t sCs;
} ttsk_gemm_desc;

int ttsk_gemm(const ttsk_gemm_desc* d, void* stream);

/* sum `n_slabs` fp32 slabs of `numel` elements (stride `slab_stride`) into dst; accumulate != 0: dst += sum */
int ttsk_reduce_slabs(const float* slabs, int n_slabs, int64_t slab_stride, float* dst, int64_t numel,
                      int accumulate, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* TTSK_H */
