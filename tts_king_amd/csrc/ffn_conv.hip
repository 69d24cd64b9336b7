// ffn_conv.hip — window kernels for the two wide Conv1d families of the FS2 step, forward (and, through a transposed weight pack,
// input gradient):  the FFT block's first position-wise conv  h = relu(Conv1d(256 -> d_ff, k = 9)(x))  and the PostNet's
// Conv1d(512 -> 512, k = 5).
// reference: fs_two/transformer/SubLayers.py:93-101 (PositionwiseFeedForward.forward: w_1 on x^T, relu), called once per
// FFTBlock (Layers.py:25-34); fs_two/transformer/Layers.py:85-129,133-143 (PostNet convolutions).  62.6 % + 17.4 % of the step's
// FLOPs sit in these convs and their gradients.
//
// The implicit-GEMM kernels re-fetch the activation rows of a tile for every tap and synchronise their eight waves per K step
// (0.9-1.3 us per step against 0.55 us of MFMA time; 45 us for the decoder's 31.9 GFLOP).  This kernel is the HiFi-GAN window
// kernel's layout (convwin.hip) for these shapes: a workgroup owns TT frames of one utterance x 256 of the output channels; the
// activation window (TT + 8 rows, all input channels) is loaded into LDS once; each of the eight waves owns 32 output channels and
// streams its weights L2 -> registers, one step = a tap x a 128-channel part of the input = 8 fragments = 32 VGPRs, requested two
// steps ahead (three register sets); no barrier inside the tap loop.  w_1 at T = 423: 112-frame tiles, 4 x 16 x 4 = 256
// workgroups, one per CU.  PostNet: 64-frame tiles (66 KiB window of 512 channels), 7 x 16 x 2 = 224 workgroups.
// The weights come in MFMA-fragment order ([tap][k-step][cout tile][lane][8], ttsk_win_conv_pack_batch): a fragment is 1 KiB
// contiguous; from the tap-major shadow (16 rows x 64 B per fragment) the kernel only ties the GEMM (kept for Cin = 256 as the
// unpacked variant of ttsk_ffn_conv_fwd).
#include "common.h"
#include "tapring.h"

namespace {

#ifndef WC_XFIRST
#define WC_XFIRST 1
#endif
#ifndef WC_WAVES
#define WC_WAVES 8
#endif
// 256 output channels per workgroup: 8 waves x 32 channels (two MFMAs per activation fragment read, two waves per SIMD).  -DWC_WAVES=4
// builds 4 waves x 64 (four MFMAs per fragment read, one wave per SIMD, accumulators in AGPRs): measured slower, 34.9 vs 31.5 us on
// w_1's forward — the tap loop runs at the MFMA pipe's rate either way (tools/debug/wc_stamps.py), the prologue and epilogue have
// half the threads.
constexpr int WC_H = 4, WC_NW = WC_WAVES, WC_NT = WC_NW * 64, WC_KH = 4, WC_CT = 16 / WC_NW, WC_COUT = WC_NW * WC_CT * 16;

struct WcArgs {
  const bf16_t* x;      // [B*S][CIN] bf16 (PAD rows are zeros)
  const bf16_t* w;      // fragment-major pack, or (Cout, K, CIN) tap-major when !PACKED
  const float* bias;    // [Cout] or null
  void* out;            // [B*S][Cout] bf16, or fp32 when OUT32
  int S, K, Cout, relu;
  int B, tiles_per_utt;
  const bf16_t* gate;   // [B*S][Cout] bf16 or null: out = gate > 0 ? out : 0 (a ReLU's backward on the way out; bf16 output only)
  // Contraction split over the input channels (an input gradient whose "input" is wide: w_1's dX has 1024, q|k|v's 768): workgroup
  // group `sp` of nsplit reads channels [sp*CIN, (sp+1)*CIN) of rows `ldx` = nsplit*CIN elements long, the k-steps
  // sp*CIN/32 .. of every tap of the pack (whose taps hold ldx/32 k-steps) and writes its own fp32 slab (out + sp*out_split): the raw
  // split-K slabs ttsk_layernorm_bwd_slabs sums.  nsplit = 1: ldx = CIN.
  int nsplit, ldx;
  int64_t out_split;
  // attention backward's delta folded into the fc input-gradient conv (bf16 output, Cout = H*128): delta[(b*H + h)*S + t] =
  // sum over head h's 128 columns of out[t][.] (as stored, bf16) * o32[t][.]  — what flash_delta_kernel computes from the same operands
  const float* o32;     // [B*S][Cout] fp32 or null
  float* delta;         // [B*H][S]
  // BatchNorm statistics of the output folded into the conv that produces it (fp32 output, CIN = 512: the PostNet's 512 -> 512 convs):
  // stats[tile][2*Cout] = per-channel sum | sum of squares over the tile's rows that exist (t < S, t < frame_limit[0]) — the partial
  // rows ttsk_bn_train_apply sums, one per (utterance, 64-frame tile), instead of a ttsk_bn_stats_slab launch over the stored rows
  float* stats;
  const int* frame_limit;
  // Stride-8 ConvTranspose1d (k = 16, padding 4: HiFi-GAN's first two upsamplers) as this conv with K = 2 pseudo-taps over the input
  // frames: output frame 8t + r takes x[t] (slot 0: weight tap r + 4) and x[t - 1] (r < 4: tap r + 12) or x[t + 1] (r >= 4: tap r - 4),
  // so with Cout = 8 * C_out "channels" (phase-major) the output rows [B*T][8 * C_out] ARE the (B, 8T, C_out) tensor.  ups_cout =
  // C_out switches slot 1's row shift to that rule (a channel group lies inside one phase: C_out >= the group's width).
  int ups_cout;
  int ups_half;       // phases r < ups_half take x[t - 1] on slot 1, the others x[t + 1] (stride / 2: 4 for stride 8; 1 for stride 2, k = 4, padding 1)
  // BatchNorm backward statistics of the layer BELOW folded into the input-gradient conv that produces its upstream gradient (bf16
  // output, CIN = 512: the PostNet's 512 -> 512 convs, Layers.py:133-143 backwards): with this conv's output tile = dL/d(dropout(tanh(
  // BN(yc)))) in hand, bnb_stats[tile][2*Cout] = per-channel sum of dy | sum of dy * xhat over the tile's rows that exist, where xhat =
  // (yc - mean) * rstd and dy = out * keep * 1/(1-p) * (1 - tanh^2(gamma * xhat + beta)) — the partial rows ttsk_bn_bwd_apply_slab sums,
  // instead of a ttsk_bn_bwd_stats_slab launch re-reading out and yc (csrc/batchnorm.hip:bn_dy is the same arithmetic).
  const float* bnb_x;        // yc [B*S][Cout] fp32, or null: no statistics
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  const uint8_t* bnb_keep;   // the forward's dropout keep bits, one byte per (row, channel quad); may be null when bnb_p == 0
  float* bnb_stats;
  float bnb_p;
  int bnb_tanh;
  // The PostNet's ends (round 5): Conv1d(80 -> 512) and Conv1d(512 -> 80), k = 5, and their input gradients (Layers.py:85-129).
  // cin_valid: the rows of x hold this many channels (ldx of them per row); the window and the pack are padded with zeros to CIN = 96
  // (0: CIN itself).  resid32 [B*S][Cout] fp32 or null: added to the output before it is rounded (bf16 output: the PostNet's first
  // conv's input gradient + the mel loss's own gradient, fastspeech2.py:104).
  int cin_valid;
  const float* resid32;
  bf16_t* out16;        // fp32 output only: a bf16 copy of the rows as well, or null (mel_linear: the mel for the loss + the PostNet's input)
  float lrelu_slope;    // > 0 (and relu == 0): out = max(v, slope * v) — HiFi-GAN's conv_pre, whose only reader is the first upsampler's LeakyReLU
  // round 6: the plain 256-channel instances of 112-frame tiles (w_1's forward and input gradient: 13 of the step's launches) fetch the rows of their A
  // operand permuted — row l15 of tile cc = channel 32 * wave + 8 * (l15 >> 2) + 4 * cc + (l15 & 3) of the group — so that a lane's 4 + 4 accumulator
  // rows are 8 consecutive channels of one frame, and every wave stores its results straight from the registers when ITS taps are done: no barrier
  // behind the tap loop (the wave in slot 0 of a SIMD used to wait 8 us there for the one in slot 1), no staging tile, no copy-out.  Same products in the
  // same order: bit-identical.  Set by the launcher when nothing else rides on the staged tile (gate, delta, statistics, bf16 copy).
  int direct;
#ifdef TTSK_STAMPS
  unsigned long long* stamps;   // diagnostic build only (make stamps; ttsk_win_conv_set_stamps): 24 slots per workgroup
#endif
};
#ifdef TTSK_STAMPS
#define WC_STAMP(i)                                                                                              \
  do {                                                                                                           \
    if (a.stamps && threadIdx.x == 0) {                                                                          \
      a.stamps[(int64_t)blockIdx.x * 24 + (i)] = __builtin_amdgcn_s_memrealtime();                               \
      if ((i) == 2 || (i) == 3) a.stamps[(int64_t)blockIdx.x * 24 + 4 + (i)] = __builtin_amdgcn_s_memtime();     \
    }                                                                                                            \
  } while (0)        // slots 6, 7: the shader clock's counter on either side of the tap loop (in-kernel clock = its delta / the 100 MHz delta)
#else
#define WC_STAMP(i) do {} while (0)        // the product library carries no stamp code and no global state for it
#endif

// Weight packs.  src = storage (Cs, K, Ds) bf16 tap-major.
//   transpose = 0: the conv's own weights, W'[co][tap][ci] = src[co][tap][ci]                   (Cout' = Cs, Cin' = Ds)
//   transpose = 1: the weights of its input gradient seen as a forward conv on dy, taps flipped:
//                  W'[co' = ci][tap][ci' = co] = src[co][K-1-tap][ci]                            (Cout' = Ds, Cin' = Cs)
// dst = [K][Cin'/32][Cout'/16][64][8]: lane l of a fragment holds W'[c*16 + (l & 15)][tap][ks*32 + (l >> 4)*8 + j], j = 0..7.
// One workgroup row (blockIdx.y) per item; the item table lives in device memory (the model's packs have fixed addresses, so it
// is built once), any number of items per launch.
__device__ __forceinline__ void pack_one(const ttsk_pack_item& it) {
  const bf16_t* __restrict__ src = (const bf16_t*)it.src;
  bf16_t* __restrict__ dst = (bf16_t*)it.dst;
  const int Cs = it.Cs, K = it.K, Ds = it.Ds;
  if ((it.transpose ? Cs : Ds) % 32 != 0) {
    // contraction not a multiple of 32 (the PostNet's 80 mel channels): k-steps padded with zeros, element by element (small weights)
    const int Co = it.transpose ? Ds : Cs, Ci = it.transpose ? Cs : Ds, nks = (Ci + 31) / 32;
    const int64_t n = (int64_t)K * nks * (Co / 16) * 512;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
      int64_t f = i >> 9;
      const int c = (int)(f % (Co / 16)); f /= (Co / 16);
      const int ks = (int)(f % nks);
      const int tap = (int)(f / nks);
      const int co = c * 16 + (l & 15), ci = ks * 32 + (l >> 4) * 8 + j;
      bf16_t v = 0;
      if (ci < Ci) v = it.transpose ? src[((int64_t)ci * K + (K - 1 - tap)) * Ds + co] : src[((int64_t)co * K + tap) * Ds + ci];
      dst[i] = v;
    }
  } else if (!it.transpose) {
    const int Co = Cs, Ci = Ds;
    const int64_t n8 = (int64_t)Co * K * (Ci / 8);                 // 16-byte pieces of the pack
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
      const int l = (int)(i & 63);
      int64_t f = i >> 6;
      const int c = (int)(f % (Co / 16)); f /= (Co / 16);
      const int ks = (int)(f % (Ci / 32));
      const int tap = (int)(f / (Ci / 32));
      const int co = c * 16 + (l & 15), ci = ks * 32 + (l >> 4) * 8;
      *(uint4*)(dst + i * 8) = *(const uint4*)(src + ((int64_t)co * K + tap) * Ds + ci);
    }
  } else if (Ds % 256 == 0 && Cs % 32 == 0) {
    // W'[co' = ci_s][tap][ci' = co_s] = src[co_s][K-1-tap][ci_s], through LDS: a tile = one k-step (32 storage rows co_s) x 256 storage
    // columns ci_s = 16 whole fragments.  Rows are read as 512-byte runs, fragments are written as 1 KiB runs (the register-transposing
    // path below stores 16-byte pieces 128 bytes apart: eight partial writes per line, and this launch is HBM-bound).
    __shared__ __attribute__((aligned(16))) unsigned short tile[32][256 + 8];
    const int Co = Ds, Ci = Cs;
    const int ncb = Co / 256, nks = Ci / 32;
    const int ntile = K * nks * ncb;
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int cb = t % ncb, ks = (t / ncb) % nks, tap = t / (ncb * nks);
      __syncthreads();                                       // the previous tile's readers are done
#pragma unroll
      for (int it = 0; it < 4; ++it) {                       // 32 rows x 32 chunks of 16 bytes
        const int idx = it * 256 + tid, r = idx >> 5, ch = idx & 31;
        *(uint4*)&tile[r][ch * 8] = *(const uint4*)(src + ((int64_t)(ks * 32 + r) * K + (K - 1 - tap)) * Ds + cb * 256 + ch * 8);
      }
      __syncthreads();
      bf16_t* out = dst + ((int64_t)(tap * nks + ks) * (Co / 16) + cb * 16) * 512;       // 16 fragments, contiguous
#pragma unroll
      for (int it = 0; it < 4; ++it) {                       // 16 fragments x 64 pieces
        const int pi = it * 256 + tid, c = pi >> 6, l = pi & 63;
        const int col = c * 16 + (l & 15), r0 = (l >> 4) * 8;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[r0 + j][col];
        *(uint4*)(out + (int64_t)pi * 8) = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
                                                      v[6] | ((unsigned)v[7] << 16));
      }
    }
  } else {
    // W'[co' = ci_s][tap][ci' = co_s] = src[co_s][K-1-tap][ci_s]: a thread takes an 8 x 8 block (8 storage rows co_s = 8 consecutive
    // ci' of one piece, 8 consecutive ci_s = co' of 8 neighbouring lanes' pieces): eight 16-byte loads, transposed in registers, eight
    // 16-byte stores
    const int Co = Ds, Ci = Cs;
    const int64_t nb = (int64_t)(Co / 8) * K * (Ci / 8);
    for (int64_t b = blockIdx.x * 256ll + threadIdx.x; b < nb; b += (int64_t)gridDim.x * 256) {
      const int co8 = (int)(b % (Co / 8));
      int64_t f = b / (Co / 8);
      const int ci8 = (int)(f % (Ci / 8));
      const int tap = (int)(f / (Ci / 8));
      uint4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = *(const uint4*)(src + ((int64_t)(ci8 * 8 + j) * K + (K - 1 - tap)) * Ds + co8 * 8);
      const int ks = (ci8 * 8) / 32, lhi = ((ci8 * 8) % 32) / 8;
#pragma unroll
      for (int m = 0; m < 8; ++m) {            // output piece of co' = co8*8 + m: elements j = r[j] half m
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned w = m < 2 ? r[j].x : (m < 4 ? r[j].y : (m < 6 ? r[j].z : r[j].w));
          v[j] = (unsigned short)((m & 1) ? (w >> 16) : (w & 0xFFFFu));
        }
        const int co = co8 * 8 + m;
        const int64_t piece = ((int64_t)(tap * (Ci / 32) + ks) * (Co / 16) + co / 16) * 64 + lhi * 16 + (co & 15);
        *(uint4*)(dst + piece * 8) = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
                                                v[6] | ((unsigned)v[7] << 16));
      }
    }
  }
}

struct ItemChunk {
  ttsk_pack_item it[48];
};
__global__ __launch_bounds__(256) void win_pack_kernel(const ttsk_pack_item* __restrict__ items) { pack_one(items[blockIdx.y]); }
__global__ __launch_bounds__(256) void win_pack_args_kernel(const ItemChunk c) { pack_one(c.it[blockIdx.y]); }

// NWV waves x CTV cout tiles of 16: the workgroup's channel group.  8 x 2 = 256 channels everywhere but on the phoneme side, where a
// 1,024-row problem is 16 tiles x Cout / 256 = 48-64 workgroups that each stream 1.2 MB of weights through one CU's L2 port (22 us
// for w_1's 4.8 GFLOP): 4 x 1 = 64 channels there — 192-256 workgroups, 0.3 MB each.
// (HiFi-GAN's 128 -> 64 upsampler — four waves, 67 KiB of LDS — is held to 256 registers so that TWO workgroups share a CU and one's window
// load / stores run under the other's taps: with the default bound the compiler took 316 and a CU held one: 46 us per launch.)
template <int CIN, int TT, bool OUT32, bool PACKED, int NWV = WC_NW, int CTV = WC_CT, bool F16 = false>
__global__ __launch_bounds__(NWV * 64, (F16 && NWV == 4 && CIN == 128) ? 2 : 1) void win_conv_kernel(const WcArgs a) {
  constexpr int COUT = NWV * CTV * 16;
  // a step = one tap x KH k-steps of 32 input channels: 128-channel parts, or the whole (padded) contraction when it is shorter
  constexpr int C = CIN, RS = CIN * 2 + 32, NT = NWV * 64, CH8 = C / 8, KH = CIN % 128 == 0 ? WC_KH : CIN / 32, CT = CTV, NF = TT / 16,
                NP = CIN / (KH * 32);
  static_assert(CIN % 32 == 0 && NP * KH * 32 == CIN, "CIN = parts of KH k-steps");
  constexpr int XROWS = TT + 2 * WC_H;
  constexpr int SRS = OUT32 ? COUT * 4 + 32 : (RS > COUT * 2 + 32 ? RS : COUT * 2 + 32);      // row stride of the output staging tile (fp32 rows are 1 KiB; bf16 rows reuse the window's stride unless the window is narrower than the output: CIN = 96)
  // (the instances that can emit BatchNorm statistics — the PostNet's: 512 or 96 input channels, fp32 output — keep NH - 1 rows of partial
  // sums behind the staged output tile)
  constexpr bool STATS = OUT32 && (CIN == 512 || CIN == 96);
  constexpr int NH = NT / COUT;                                   // row groups of the tile in that epilogue: 512 threads x 256 channels: two; 320 x 80: four
  constexpr int STG = TT * SRS + (STATS ? (NH - 1) * 2 * COUT * 4 : 0);
  constexpr int SMEM = XROWS * RS > STG ? XROWS * RS : STG;
  __shared__ __attribute__((aligned(16))) unsigned char XW[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  // XCD-aware mapping: consecutive workgroup ids go round the 8 XCDs, and a channel group's weights (1.2 MB at d_ff = 1024, k = 9) should
  // stay in ONE XCD's 4 MiB L2 instead of all groups in every L2: channel group = f(id % 8), tile = the rest.
  int bi, t0, cg, sp;
  {
    const int id = blockIdx.x, ncg = (a.Cout / COUT) * a.nsplit, ntile = a.tiles_per_utt * a.B;
    const int xcd = id & 7, per = 8 / (ncg < 8 ? ncg : 8);        // XCDs per channel group (ncg = 1, 2, 4, 8); other counts: plain order
    int tile;
    if ((8 % (ncg < 8 ? ncg : 8)) == 0 && ncg <= 8 && (ntile * ncg) % 8 == 0 && ntile % per == 0) {
      cg = xcd / per;
      tile = (id >> 3) * per + (xcd % per);
    } else if (ncg % 8 == 0) {                                      // ncg / 8 channel groups per XCD
      const int gpx = ncg >> 3, slot = id >> 3;
      tile = slot / gpx;
      cg = xcd * gpx + (slot - tile * gpx);
    } else {
      cg = id / ntile;
      tile = id - cg * ntile;
    }
    bi = tile / a.tiles_per_utt;
    t0 = (tile - bi * a.tiles_per_utt) * TT;
    sp = cg % a.nsplit;            // group = (channel group, contraction split)
    cg = cg / a.nsplit;
  }
  WC_STAMP(0);
  const int S = a.S, K = a.K, HK = (K - 1) / 2, NS = NP * K;       // NS steps: tap g / NP, 128-channel part g % NP
  const int ldx = a.ldx;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * S * ldx + sp * C;
  const bf16_t* __restrict__ wbase = a.w;
  const int ks_total = (ldx + 31) / 32, ks_base = sp * (C / 32);
  const int cin_valid = a.cin_valid > 0 ? a.cin_valid : C;

  // weight fragments by buffer loads (tapring.h): one per-lane byte offset for the whole kernel, the step's distance in a scalar register
  const __amdgpu_buffer_rsrc_t wres = weights_rsrc(wbase, K * ks_total * 32 * a.Cout * 2);
  const int wlane = PACKED ? (cg * (COUT / 16) + wave * CT) * 1024 + lane * 16 : ((cg * COUT + wave * CT * 16 + l15) * K * C + q * 8) * 2;
  constexpr bool CAN_DIRECT = PACKED && !F16 && CIN == 256 && TT == 112 && NWV == 8 && CTV == 2;
  const bool direct = CAN_DIRECT && a.direct;
  int wl[CT];                                                    // PACKED: this lane's byte offset into a (tap, k-step) of the pack, per cout tile
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int co = 32 * wave + 8 * (l15 >> 2) + 4 * cc + (l15 & 3);
    wl[cc] = direct ? (cg * (COUT / 16) + (co >> 4)) * 1024 + ((co & 15) + 16 * q) * 16 : wlane + cc * 1024;
  }
  const int kstep_bytes = (a.Cout / 16) * 1024;                  // PACKED: bytes per (tap, k-step)
  bf16x8 wa[KH][CT], wb[KH][CT], wc[KH][CT];      // three register sets: a step's weights are requested two steps (>= 1 us) ahead
  auto load_w = [&](int g, bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
    const int tap = g / NP, part = g - tap * NP;
    if (PACKED) {
      frags_load<KH, CT>(w, wres, wl, (tap * ks_total + ks_base + part * KH) * kstep_bytes, kstep_bytes);
    } else {
      const int soff = (tap * C + part * (KH * 32)) * 2;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) w[ks][cc] = frag_load(wres, wlane, soff + ks * 64 + cc * (16 * K * C * 2));
    }
  };
#if !WC_XFIRST
  load_w(0, wa);
  if (!F16 || 1 < NS) load_w(min(1, NS - 1), wb);
  if (!F16 || 2 < NS) load_w(min(2, NS - 1), wc);
#endif

  {  // ---- activation window: rows t0 - 4 .. t0 + TT + 4 of the utterance, zeros outside it (the conv's zero padding).
     // Requested BEFORE the weight fragments: loads return in order, nothing starts before the window is in LDS, and behind three
     // register sets of fragments (24 KiB per wave) it would arrive last.
    constexpr int NCH = (XROWS * CH8 + NT - 1) / NT;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - WC_H + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < XROWS * CH8 && t >= 0 && t < S && ch * 8 < cin_valid) xv[it] = *(const uint4*)(xb + (int64_t)t * ldx + ch * 8);
    }
#if WC_XFIRST
    load_w(0, wa);
    if (!F16 || 1 < NS) load_w(min(1, NS - 1), wb);      // (fp16 instances: no repeated requests, see the tap loop)
    if (!F16 || 2 < NS) load_w(min(2, NS - 1), wc);
#endif
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < XROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = xv[it];
    }
  }
  f32x4 bv[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
    bv[cc] = a.bias ? *(const f32x4*)(a.bias + cg * COUT + (direct ? 32 * wave + 8 * q + 4 * cc : (wave * CT + cc) * 16 + q * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
  WC_STAMP(1);
  __syncthreads();
  WC_STAMP(2);

  f32x4 acc[CT][NF];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const int inl = (l15 + WC_H) * RS + q * 16;               // this lane's window position at tap shift 0 (byte offset into XW)
    // row shift of tap slot 1 in the stride-8 transposed-conv mode (slot 0: none)
    // (only HiFi-GAN's fp16 instances have that mode: the training instances carry no test for it in their step code)
    const int ups_shift = F16 && a.ups_cout ? ((cg * COUT + wave * CT * 16) / a.ups_cout < a.ups_half ? -1 : 1) : 0;      // a wave's channels lie in one phase
    // Activation fragments through a ring of NF registers, refilled NF * CT MFMAs ahead of their use (tapring.h).  14 tiles — the
    // HiFi-GAN upsamplers' 224-frame tiles — hold 112 accumulator + 96 weight registers: no room for a ring, the fragments are read
    // where they are used.
    constexpr bool RING = NF <= 8;
    auto inp_of = [&](int g) __attribute__((always_inline)) {
      const int tap = g / NP, part = g - tap * NP;
      return inl + (F16 && a.ups_cout ? tap * ups_shift : tap - HK) * RS + part * (KH * 64);
    };
    bf16x8 ring[RING ? NF : 1];
    if constexpr (RING) ring_prime<NF, RS>(ring, XW, inp_of(0));
    auto step = [&](int g, const bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
      const int inp = inp_of(g);
      if constexpr (RING) {
        tap_ring<F16, KH, CT, NF, RS>(acc, ring, w, XW, inp, inp_of(g + 1 < NS ? g + 1 : g));
      } else {
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            const bf16x8 Bf = *(const bf16x8*)(XW + inp + i * 16 * RS + ks * 64);
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) acc[cc][i] = mfma16<F16>(w[ks][cc], Bf, acc[cc][i]);
          }
        }
      }
    };
    // the weight requests are unconditional (past the last step they fetch its fragments again, into registers nobody reads): exact wait
    // counts.  Not in the fp16 instances: HiFi-GAN's upsamplers have 2-8 steps, the repeats would be half their weight traffic (the
    // 128 -> 64 upsampler measured 47 -> 52 us with them).
    constexpr bool UNCOND = !F16;
#pragma unroll 1
    for (int g = 0; g < NS; g += 3) {
      step(g, wa);
      if (UNCOND || g + 3 < NS) load_w(min(g + 3, NS - 1), wa);
      if (g + 1 < NS) step(g + 1, wb);
      if (UNCOND || g + 4 < NS) load_w(min(g + 4, NS - 1), wb);
      if (g + 2 < NS) step(g + 2, wc);
      if (UNCOND || g + 5 < NS) load_w(min(g + 5, NS - 1), wc);
    }
  }
  WC_STAMP(3);
#ifdef TTSK_STAMPS
  if (a.stamps && lane == 0 && wave < 8) {
    a.stamps[(int64_t)blockIdx.x * 24 + 8 + wave] = __builtin_amdgcn_s_memrealtime();      // slots 8..15: each wave's end of the tap loop
    a.stamps[(int64_t)blockIdx.x * 24 + 16 + wave] = __builtin_amdgcn_s_getreg(63492);      // slots 16..23: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], ...)
  }
#endif
  if constexpr (CAN_DIRECT) {
    if (direct) {        // this wave's 32 channels x TT frames from its registers to memory, 16 (bf16) / 2 x 16 (fp32) bytes per frame tile and lane
      const int co8 = cg * COUT + 32 * wave + 8 * q;
      const int64_t row0 = (int64_t)bi * S;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int t = t0 + i * 16 + l15;
        f32x4 v[2] = {acc[0][i] + bv[0], acc[1][i] + bv[1]};
        if (a.relu) {
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[cc][e] = fmaxf(v[cc][e], 0.f);
        }
        if (t < S) {
          if constexpr (OUT32) {
            float* p = (float*)a.out + sp * a.out_split + (row0 + t) * a.Cout + co8;
            *(f32x4*)p = v[0];
            *(f32x4*)(p + 4) = v[1];
          } else {
            if (a.resid32) {        // (the fp32 residual joins the accumulators: the sum is rounded once)
              const float* rp = a.resid32 + (row0 + t) * a.Cout + co8;
              v[0] += *(const f32x4*)rp;
              v[1] += *(const f32x4*)(rp + 4);
            }
            *(uint4*)((bf16_t*)a.out + sp * a.out_split + (row0 + t) * a.Cout + co8) =
                make_uint4(pack2<F16>(v[0][0], v[0][1]), pack2<F16>(v[0][2], v[0][3]), pack2<F16>(v[1][0], v[1][1]), pack2<F16>(v[1][2], v[1][3]));
          }
        }
      }
      WC_STAMP(5);
      return;
    }
  }
  // (BatchNorm-backward statistics, see the end of the kernel: the layer-below rows this thread will need are requested now, so that
  // they arrive during the staging and the stores)
  f32x4 bnb_v[8];
  unsigned bnb_kb[8];
  if constexpr (!OUT32 && (CIN == 512 || CIN == 96) && !F16 && NWV == 8 && CTV == 2) {
    if (a.bnb_x) {
      const int cq = cg * COUT + (tid & 63) * 4, rg = tid >> 6;
#pragma unroll
      for (int k = 0; k < 8; ++k) {                 // rows past the utterance: clamped here, zeroed where they are used
        const int t = min(t0 + rg + 8 * k, S - 1);
        const int64_t row = (int64_t)bi * S + t;
        bnb_v[k] = *(const f32x4*)(a.bnb_x + row * a.Cout + cq);
        bnb_kb[k] = (a.bnb_p > 0.f && a.bnb_keep) ? a.bnb_keep[row * (a.Cout >> 2) + (cq >> 2)] : 0xFu;
      }
    }
  }
  __syncthreads();          // every wave is done with the window: its rows become the output staging tile
  WC_STAMP(4);

#pragma unroll
  for (int i = 0; i < NF; ++i) {
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      f32x4 v = acc[cc][i] + bv[cc];
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (F16 && a.lrelu_slope > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.lrelu_slope);
      }
      const int col = (wave * CT + cc) * 16 + q * 4;
      if (!OUT32 && a.resid32) {        // the fp32 residual joins the accumulators: the sum is rounded ONCE, when it is packed below
        const int t = t0 + i * 16 + l15;
        if (t < S) v += *(const f32x4*)(a.resid32 + ((int64_t)bi * S + t) * a.Cout + cg * COUT + col);
      }
      if (OUT32) *(f32x4*)(XW + (i * 16 + l15) * SRS + col * 4) = v;
      else *(uint2*)(XW + (i * 16 + l15) * SRS + col * 2) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
    }
  }
  __syncthreads();
  if constexpr (STATS) {
    static_assert(NT % COUT == 0 && TT % NH == 0 && SMEM >= TT * SRS + (NH - 1) * 2 * COUT * 4, "room for the partial sums behind the staging tile");
    if (a.stats) {
      const int half = tid / COUT, c = tid - half * COUT;
      const int lim = a.frame_limit ? (a.frame_limit[0] < S ? a.frame_limit[0] : S) : S;
      float sm = 0.f, sq = 0.f;
      for (int r = half * (TT / NH); r < (half + 1) * (TT / NH); ++r) {
        if (t0 + r < lim) { const float v = *(const float*)(XW + r * SRS + c * 4); sm += v; sq += v * v; }
      }
      float* red = (float*)(XW + TT * SRS);                       // [NH - 1][2][COUT]
      if (half > 0) { red[((half - 1) * 2 + 0) * COUT + c] = sm; red[((half - 1) * 2 + 1) * COUT + c] = sq; }
      if (NH > 1) __syncthreads();
      if (half == 0) {
#pragma unroll
        for (int h = 1; h < NH; ++h) { sm += red[((h - 1) * 2 + 0) * COUT + c]; sq += red[((h - 1) * 2 + 1) * COUT + c]; }
        float* P = a.stats + ((int64_t)bi * a.tiles_per_utt + t0 / TT) * 2 * a.Cout + cg * COUT + c;
        P[0] = sm;
        P[a.Cout] = sq;
      }
    }
  }
  constexpr int ESZ = OUT32 ? 4 : 2;
  constexpr int OCH = COUT * ESZ / 16;                // 16-byte chunks per output row of this channel group
  constexpr int NCO = (TT * OCH + NT - 1) / NT;
  unsigned char* __restrict__ ob = (unsigned char*)a.out + (sp * a.out_split + (int64_t)bi * S * a.Cout + cg * COUT) * ESZ;
#pragma unroll
  for (int it = 0; it < NCO; ++it) {
    const int idx = it * NT + tid;
    const int rr = idx / OCH, ch = idx - rr * OCH;
    const int t = t0 + rr;
    float dacc = 0.f;
    if (idx < TT * OCH && t < S) {
      uint4 v = *(const uint4*)(XW + rr * SRS + ch * 16);
      if (!OUT32 && a.gate) {
        const uint4 g = *(const uint4*)(a.gate + ((int64_t)bi * S + t) * a.Cout + cg * COUT + ch * 8);
        auto keep = [](unsigned w) {      // 0xFFFF per bf16 half that is > 0 (sign clear, not zero)
          const unsigned lo = w & 0xFFFFu, hi = w >> 16;
          return ((lo - 1u) < 0x7FFFu ? 0xFFFFu : 0u) | ((hi - 1u) < 0x7FFFu ? 0xFFFF0000u : 0u);
        };
        v.x &= keep(g.x); v.y &= keep(g.y); v.z &= keep(g.z); v.w &= keep(g.w);
      }
      *(uint4*)(ob + (int64_t)t * a.Cout * ESZ + ch * 16) = v;
      if (OUT32 && a.out16)
        *(uint2*)(a.out16 + ((int64_t)bi * S + t) * a.Cout + cg * COUT + ch * 4) =
            make_uint2(pack_bf2(__uint_as_float(v.x), __uint_as_float(v.y)), pack_bf2(__uint_as_float(v.z), __uint_as_float(v.w)));
      if (!OUT32 && a.delta) {
        const float* op = a.o32 + ((int64_t)bi * S + t) * a.Cout + cg * COUT + ch * 8;
        const f32x4 y0 = *(const f32x4*)op, y1 = *(const f32x4*)(op + 4);
        dacc = __uint_as_float(v.x << 16) * y0[0] + __uint_as_float(v.x & 0xFFFF0000u) * y0[1] + __uint_as_float(v.y << 16) * y0[2] +
               __uint_as_float(v.y & 0xFFFF0000u) * y0[3] + __uint_as_float(v.z << 16) * y1[0] + __uint_as_float(v.z & 0xFFFF0000u) * y1[1] +
               __uint_as_float(v.w << 16) * y1[2] + __uint_as_float(v.w & 0xFFFF0000u) * y1[3];
      }
    }
    if (!OUT32 && a.delta) {          // 16 consecutive lanes hold one (row, head): 16 chunks of 8 columns
      dacc = quad16_sum(dacc);
      const int hd = (cg * COUT + ch * 8) / 128, nh = a.Cout / 128;
      if (idx < TT * OCH && t < S && (ch & 15) == 0) a.delta[((int64_t)bi * nh + hd) * S + t] = dacc;
    }
  }
  if constexpr (!OUT32 && (CIN == 512 || CIN == 96) && !F16 && NWV == 8 && CTV == 2) {
    if (a.bnb_x) {
      // thread = (channel quad of the group's 256 channels, one of 8 row groups); rows rg, rg + 8, ... of the tile
      static_assert(TT == 64 && NT == 512, "eight rows per thread");
      const int quad = tid & 63, rg = tid >> 6;
      const int C = a.Cout, cq = cg * COUT + quad * 4;
      const int lim = a.frame_limit ? (a.frame_limit[0] < S ? a.frame_limit[0] : S) : S;
      const float scale = a.bnb_p > 0.f ? 1.f / (1.f - a.bnb_p) : 1.f;
      const f32x4 m = *(const f32x4*)(a.bnb_mean + cq), rs = *(const f32x4*)(a.bnb_rstd + cq);
      const f32x4 gmm = *(const f32x4*)(a.bnb_gamma + cq), bt = *(const f32x4*)(a.bnb_beta + cq);
      float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rr = rg + 8 * k;
        const uint2 u = *(const uint2*)(XW + rr * SRS + quad * 8);      // the gradient as stored (bf16)
        float dy[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
        const bool live = t0 + rr < lim;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.bnb_p > 0.f) dy[e] = ((bnb_kb[k] >> e) & 1u) ? dy[e] * scale : 0.f;
          float xh = (bnb_v[k][e] - m[e]) * rs[e];
          if (a.bnb_tanh) { const float tt = 1.f - 2.f * __frcp_rn(__expf(2.f * (xh * gmm[e] + bt[e])) + 1.f); dy[e] *= 1.f - tt * tt; }
          if (!live) { xh = 0.f; dy[e] = 0.f; }
          sm[e] += dy[e];
          sq[e] += dy[e] * xh;
        }
      }
      __syncthreads();                               // every thread is done reading the staged tile: it becomes the reduction buffer
      float* red = (float*)XW;                       // [8 row groups][2][256]
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(rg * 2 + 0) * COUT + quad * 4 + e] = sm[e]; red[(rg * 2 + 1) * COUT + quad * 4 + e] = sq[e]; }
      __syncthreads();
      {
        const int c = tid & (COUT - 1), half = tid / COUT;           // 512 threads: sums | second sums
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) acc += red[(g * 2 + half) * COUT + c];
        a.bnb_stats[((int64_t)bi * a.tiles_per_utt + t0 / TT) * 2 * C + half * C + cg * COUT + c] = acc;
      }
    }
  }
  WC_STAMP(5);
}

// ---- HiFi-GAN's 256 -> 128 stride-8 upsampler (WcArgs::ups_cout's transposed-conv mode, fp16 rows) with the window loaded ONCE per frame
// tile and the channel groups looped inside the workgroup.  On win_conv_kernel<256, 224, ..> the launch is 448 workgroups (14 tiles x 8
// utterances x 4 groups of 256 phase-major channels), one per CU in two rounds, each 4.4 us of window load + 1.5 us barrier + 10.8 us of
// taps + 3.6 us of staging and stores with nothing beside it (tools/debug/ups_stamps.py): 49 us for 11.7 us of MFMAs.  Here a workgroup
// owns 96 frames (32 tiles x 8 utterances = 256 workgroups: one round), keeps their 98-row window (53 KiB) for all NG groups, stages
// each group's 96 x 256 outputs in a tile of its own (52 KiB) and leaves its stores in flight under the next group's taps; the weights
// (all 1 MiB per workgroup, L2 -> registers) go through four register sets, one per step (2 pseudo-taps x 2 halves of the channels),
// each refilled with the next group's same step right behind its MFMAs.
template <int CIN, int TT>
__global__ __launch_bounds__(WC_NT, 1) void ups_loop_kernel(const WcArgs a) {
  constexpr int RS = CIN * 2 + 32, NT = WC_NT, CH8 = CIN / 8, KH = WC_KH, CT = WC_CT, NF = TT / 16, NP = CIN / (KH * 32), COUT = WC_COUT, NS = 2 * NP;
  static_assert(NS == 4 && NF <= 8 && TT % 16 == 0, "four steps = four register sets; the activation ring holds NF fragments");
  constexpr int XROWS = TT + 2, SRS = COUT * 2 + 32;
  __shared__ __attribute__((aligned(16))) unsigned char XW[XROWS * RS];
  __shared__ __attribute__((aligned(16))) unsigned char ST[TT * SRS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int S = a.S, NG = a.Cout / COUT;
  const int bi = blockIdx.x / a.tiles_per_utt, t0 = (blockIdx.x - bi * a.tiles_per_utt) * TT;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * S * CIN;
  const __amdgpu_buffer_rsrc_t wres = weights_rsrc(a.w, 2 * (CIN / 32) * 32 * a.Cout * 2);
  const int kstep_bytes = (a.Cout / 16) * 1024;
  bf16x8 w0[KH][CT], w1[KH][CT], w2[KH][CT], w3[KH][CT];
  auto load_w = [&](int cg, int g, bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {      // step g = pseudo-tap g / NP, channels (g % NP) * 128 ..
    const int tap = g / NP, part = g - tap * NP;
    frags_load<KH, CT>(w, wres, (cg * (COUT / 16) + wave * CT) * 1024 + lane * 16, (tap * (CIN / 32) + part * KH) * kstep_bytes, kstep_bytes);
  };
  {  // window rows t0 - 1 .. t0 + TT (zeros outside the utterance), then the first group's four steps of weights behind them
    constexpr int NCH = (XROWS * CH8 + NT - 1) / NT;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - 1 + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < XROWS * CH8 && t >= 0 && t < S) xv[it] = *(const uint4*)(xb + (int64_t)t * CIN + ch * 8);
    }
    load_w(0, 0, w0); load_w(0, 1, w1); load_w(0, 2, w2); load_w(0, 3, w3);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < XROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = xv[it];
    }
  }
  __syncthreads();
  const int inl = (l15 + 1) * RS + q * 16;                  // this lane's window position at row shift 0
  bf16x8 ring[NF];
  ring_prime<NF, RS>(ring, XW, inl);
#pragma unroll 1
  for (int cg = 0; cg < NG; ++cg) {
    const int shift = (cg * COUT + wave * CT * 16) / a.ups_cout < a.ups_half ? -1 : 1;      // a wave's 32 channels lie in one phase
    const int nx = cg + 1 < NG ? cg + 1 : cg;                 // (past the last group: its own fragments again, into registers nobody reads — exact wait counts)
    f32x4 acc[CT][NF];                                        // starts at the bias (fp32 sums: bias + products in any order round the same way only
#pragma unroll                                                // to the last bit — the stored value is fp16)
    for (int cc = 0; cc < CT; ++cc) {
      const f32x4 bv = *(const f32x4*)(a.bias + cg * COUT + (wave * CT + cc) * 16 + q * 4);
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[cc][i] = bv;
    }
    auto inp_of = [&](int g) __attribute__((always_inline)) {
      const int tap = g / NP, part = g - tap * NP;
      return inl + tap * shift * RS + part * (KH * 64);
    };
    tap_ring<true, KH, CT, NF, RS>(acc, ring, w0, XW, inp_of(0), inp_of(1));
    load_w(nx, 0, w0);
    tap_ring<true, KH, CT, NF, RS>(acc, ring, w1, XW, inp_of(1), inp_of(2));
    load_w(nx, 1, w1);
    tap_ring<true, KH, CT, NF, RS>(acc, ring, w2, XW, inp_of(2), inp_of(3));
    load_w(nx, 2, w2);
    tap_ring<true, KH, CT, NF, RS>(acc, ring, w3, XW, inp_of(3), inl);      // (every group starts at row shift 0, channels 0 ..)
    load_w(nx, 3, w3);
    __syncthreads();          // every thread is past its reads of the previous group's staged tile
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        const f32x4 v = acc[cc][i];
        *(uint2*)(ST + (i * 16 + l15) * SRS + ((wave * CT + cc) * 16 + q * 4) * 2) = make_uint2(pack2<true>(v[0], v[1]), pack2<true>(v[2], v[3]));
      }
    __syncthreads();
    constexpr int OCH = COUT * 2 / 16, NCO = TT * OCH / NT;
    static_assert(TT * OCH % NT == 0, "whole passes of the store loop");
    unsigned char* __restrict__ ob = (unsigned char*)a.out + ((int64_t)bi * S * a.Cout + cg * COUT) * 2;
#pragma unroll
    for (int it = 0; it < NCO; ++it) {
      const int idx = it * NT + tid;
      const int rr = idx / OCH, ch = idx - rr * OCH;
      if (t0 + rr < S) *(uint4*)(ob + (int64_t)(t0 + rr) * a.Cout * 2 + ch * 16) = *(const uint4*)(ST + rr * SRS + ch * 16);
    }
  }
}

#ifdef TTSK_STAMPS
static unsigned long long* g_wc_stamps = nullptr;
// diagnostic build only (`make stamps`, tools/debug/wc_stamps.py; not declared in ttsk.h, not in the product library)
extern "C" int ttsk_win_conv_set_stamps(void* dev_buffer) {
  g_wc_stamps = (unsigned long long*)dev_buffer;
  return TTSK_OK;
}
#endif

int launch_win_conv(const WcArgs& a0, int B, int S, int Cin, int out_f32, int packed, hipStream_t s, int f16 = 0) {
  WcArgs a = a0;
#ifdef TTSK_STAMPS
  a.stamps = g_wc_stamps;
#endif
  const bool short_seq = !f16 && Cin == 256 && packed && (S <= 64 || (S > 112 && S <= 128));      // phoneme-side sequences: 64-frame tiles waste less
  const int TT = Cin == 256 && !short_seq ? 112 : 64;
  a.B = B;
  a.tiles_per_utt = (S + TT - 1) / TT;
  dim3 grid(a.tiles_per_utt * B * (a.Cout / WC_COUT) * a.nsplit);
  if (f16) {
    // HiFi-GAN's operand type: the two shapes its stride-8 upsamplers need.  Their contraction is short (2 pseudo-taps x Cin), so a
    // workgroup's time is its prologue, its weight stream (0.26 / 0.52 MB) and its stores: frame tiles twice as tall as the training
    // shapes' (224 / 128 frames) halve the number of times each is paid (stage times at B = 8, T = 384: ups1 82 us on the polyphase GEMMs, 53 at 112 frames, 50 at 224; ups0 70 / 46 / 41).
    if (Cin == 80) {         // conv_pre (80 -> 512, k = 7): the PostNet's first conv's instance on fp16 rows
      a.cin_valid = 80;
      hipLaunchKernelGGL((win_conv_kernel<96, 64, false, true, WC_NW, WC_CT, true>), grid, dim3(WC_NT), 0, s, a);
      return 0;
    }
    const int TTU = Cin == 512 ? 96 : 224;      // (512 -> 256: 96-frame tiles are 4 x 8 x 8 = 256 workgroups at 384 frames per utterance — one per CU; 128-frame tiles 192)
    a.tiles_per_utt = (S + TTU - 1) / TTU;
    if (Cin == 128) {          // the stride-2 upsampler 128 -> 64: 128 phase-major channels = one group of 4 waves x 32
      const dim3 g1(a.tiles_per_utt * B * (a.Cout / 128) * a.nsplit);
      hipLaunchKernelGGL((win_conv_kernel<128, 224, false, true, 4, 2, true>), g1, dim3(256), 0, s, a);
      return 0;
    }
    const dim3 gu(a.tiles_per_utt * B * (a.Cout / WC_COUT) * a.nsplit);
    if (Cin == 256) hipLaunchKernelGGL((win_conv_kernel<256, 224, false, true, WC_NW, WC_CT, true>), gu, dim3(WC_NT), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<512, 96, false, true, WC_NW, WC_CT, true>), gu, dim3(WC_NT), 0, s, a);
    return 0;
  }
  if (Cin == 256 && a.Cout == 80) {      // mel_linear (256 -> 80, k = 1): five waves of 16 channels over 64-frame tiles
    a.tiles_per_utt = (S + 63) / 64;
    const dim3 g5(a.tiles_per_utt * B * a.nsplit);
    if (out_f32) hipLaunchKernelGGL((win_conv_kernel<256, 64, true, true, 5, 1>), g5, dim3(320), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<256, 64, false, true, 5, 1>), g5, dim3(320), 0, s, a);
    return 0;
  }
  if (Cin == 80) {           // the PostNet's first conv (80 -> 512) / its last conv's input gradient: contraction padded to 96 channels
    a.cin_valid = 80;
    if (out_f32) hipLaunchKernelGGL((win_conv_kernel<96, 64, true, true>), grid, dim3(WC_NT), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<96, 64, false, true>), grid, dim3(WC_NT), 0, s, a);
    return 0;
  }
  if (Cin == 512 && a.Cout == 80) {      // the PostNet's last conv (512 -> 80) / its first conv's input gradient: 5 waves x 16 channels
    const dim3 g5(a.tiles_per_utt * B * a.nsplit);
    if (out_f32) hipLaunchKernelGGL((win_conv_kernel<512, 64, true, true, 5, 1>), g5, dim3(320), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<512, 64, false, true, 5, 1>), g5, dim3(320), 0, s, a);
    return 0;
  }
  if (short_seq && !a.delta && (int)grid.x <= 128) {     // few workgroups, each bound by its weight stream: 64-channel groups
    dim3 g4(grid.x * 4);
    if (out_f32) hipLaunchKernelGGL((win_conv_kernel<256, 64, true, true, 4, 1>), g4, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<256, 64, false, true, 4, 1>), g4, dim3(256), 0, s, a);
    return 0;
  }
  // (the 112-frame instances with nothing riding on the staged output tile store straight from the registers: WcArgs::direct)
  a.direct = packed && !short_seq && Cin == 256 && !a.gate && !a.delta && !a.stats && !a.bnb_x && !a.out16 && !a.ups_cout && !(a.lrelu_slope > 0.f);
  if (Cin == 256 && out_f32) {
    if (short_seq) hipLaunchKernelGGL((win_conv_kernel<256, 64, true, true>), grid, dim3(WC_NT), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<256, 112, true, true>), grid, dim3(WC_NT), 0, s, a);
  } else if (short_seq) {
    hipLaunchKernelGGL((win_conv_kernel<256, 64, false, true>), grid, dim3(WC_NT), 0, s, a);
  } else if (Cin == 256) {
    if (packed) hipLaunchKernelGGL((win_conv_kernel<256, 112, false, true>), grid, dim3(WC_NT), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<256, 112, false, false>), grid, dim3(WC_NT), 0, s, a);
  } else {
    if (out_f32) hipLaunchKernelGGL((win_conv_kernel<512, 64, true, true>), grid, dim3(WC_NT), 0, s, a);
    else hipLaunchKernelGGL((win_conv_kernel<512, 64, false, true>), grid, dim3(WC_NT), 0, s, a);
  }
  return 0;
}

}  // namespace

extern "C" int ttsk_win_conv_supported(int Cin, int Cout, int K) {
  if (!(K >= 1 && K <= 2 * WC_H + 1 && (K & 1) == 1)) return 0;
  if ((Cin == 256 || Cin == 512 || Cin == 80) && Cout > 0 && Cout % WC_COUT == 0) return 1;
  return (Cin == 512 || Cin == 256) && Cout == 80;          // (80 output channels: the PostNet's last conv / first conv's input gradient; mel_linear)
}
extern "C" int ttsk_ffn_conv_supported(int Cin, int Cout, int K) { return Cin == 256 && ttsk_win_conv_supported(Cin, Cout, K); }

extern "C" int ttsk_win_conv_pack_table(const ttsk_pack_item* dev_items, int n, void* stream) {
  TTSK_REQUIRE(dev_items && n > 0 && n <= 65535 && (((uintptr_t)dev_items) & 7) == 0, "ttsk_win_conv_pack_table: bad arguments");
  hipLaunchKernelGGL(win_pack_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, dev_items);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// host items: checked, handed to the kernel through its arguments (capturable, no device table needed)
extern "C" int ttsk_win_conv_pack_items(const ttsk_pack_item* items, int n, void* stream) {
  TTSK_REQUIRE(items && n > 0 && n <= 48, "ttsk_win_conv_pack_items: 1..48 items");
  ItemChunk c;
  for (int i = 0; i < 48; ++i) {
    const ttsk_pack_item& it = items[i < n ? i : 0];
    TTSK_REQUIRE(it.src && it.dst && ((((uintptr_t)it.src) | ((uintptr_t)it.dst)) & 15) == 0, "ttsk_win_conv_pack_items: null / unaligned pointer");
    TTSK_REQUIRE(it.Cs > 0 && it.Cs % 16 == 0 && it.Ds > 0 && it.Ds % 16 == 0 && it.K >= 1, "ttsk_win_conv_pack_items: bad shape (%d, %d, %d)", it.Cs, it.K, it.Ds);
    c.it[i] = it;
  }
  hipLaunchKernelGGL(win_pack_args_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, c);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_win_conv_pack_batch(const void* const* w_bf16, void* const* packed_bf16, int n, int Cs, int K, int Ds, int transpose,
                                        void* stream) {
  TTSK_REQUIRE(w_bf16 && packed_bf16 && n > 0 && n <= 16, "ttsk_win_conv_pack_batch: bad arguments");
  ttsk_pack_item items[16];
  for (int i = 0; i < n; ++i) items[i] = ttsk_pack_item{w_bf16[i], packed_bf16[i], Cs, K, Ds, transpose};
  return ttsk_win_conv_pack_items(items, n, stream);
}
extern "C" int ttsk_ffn_pack_weight_batch(const void* const* w_bf16, void* const* packed_bf16, int n, int Cout, int K, void* stream) {
  return ttsk_win_conv_pack_batch(w_bf16, packed_bf16, n, Cout, K, 256, 0, stream);
}
extern "C" int ttsk_ffn_pack_weight(const void* w_bf16, void* packed_bf16, int Cout, int K, void* stream) {
  return ttsk_win_conv_pack_batch(&w_bf16, &packed_bf16, 1, Cout, K, 256, 0, stream);
}

extern "C" int ttsk_win_conv(const void* x_bf16, const void* w_packed, const float* bias, const void* gate_bf16, const float* delta_o32,
                             float* delta_out, void* out, int out_f32, int B, int S, int Cin, int Cout, int K, int relu, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_packed && out, "ttsk_win_conv: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_win_conv: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE(ttsk_win_conv_supported(Cin, Cout, K), "ttsk_win_conv: no instance for Cin=%d Cout=%d K=%d", Cin, Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_packed) | ((uintptr_t)bias) | ((uintptr_t)out)) & 15) == 0, "ttsk_win_conv: 16-byte alignment");
  TTSK_REQUIRE((int64_t)B * S * (Cout > Cin ? Cout : Cin) * 4 < ((int64_t)1 << 40), "ttsk_win_conv: sizes out of range");
  TTSK_REQUIRE(!(gate_bf16 && out_f32) && (((uintptr_t)gate_bf16) & 15) == 0, "ttsk_win_conv: the gate goes with bf16 output, 16-byte aligned");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_packed, bias, out, S, K, Cout, relu, 0, 0, (const bf16_t*)gate_bf16, 1, Cin, 0, delta_o32, delta_out};
  TTSK_REQUIRE(!delta_out || (delta_o32 && !out_f32 && Cout % 128 == 0 && (((uintptr_t)delta_o32) & 15) == 0),
               "ttsk_win_conv: delta needs o32 (16-byte aligned), bf16 output and Cout = heads * 128");
  launch_win_conv(a, B, S, Cin, out_f32, 1, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// ttsk_win_conv (bf16 output, no bias / gate) + an fp32 residual added before the rounding: the PostNet's first conv's input gradient
// (512 -> 80 on the transposed pack) + the mel loss's own gradient (fastspeech2.py:104: postnet(output) + output)
extern "C" int ttsk_win_conv_resid(const void* x_bf16, const void* w_packed, const float* resid_f32, void* out_bf16, int B, int S, int Cin,
                                   int Cout, int K, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_packed && resid_f32 && out_bf16, "ttsk_win_conv_resid: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_win_conv_resid: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE(ttsk_win_conv_supported(Cin, Cout, K), "ttsk_win_conv_resid: no instance for Cin=%d Cout=%d K=%d", Cin, Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_packed) | ((uintptr_t)resid_f32) | ((uintptr_t)out_bf16)) & 15) == 0, "ttsk_win_conv_resid: 16-byte alignment");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_packed, nullptr, out_bf16, S, K, Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr};
  a.resid32 = resid_f32;
  launch_win_conv(a, B, S, Cin, 0, 1, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// ttsk_win_conv with fp32 output AND a bf16 copy of it (mel_linear: fastspeech2.py:102 — the fp32 mel goes to the loss and is added back
// behind the PostNet, the bf16 copy is the PostNet's input)
extern "C" int ttsk_win_conv_dual(const void* x_bf16, const void* w_packed, const float* bias, float* out_f32, void* out_bf16, int B, int S,
                                  int Cin, int Cout, int K, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_packed && out_f32 && out_bf16, "ttsk_win_conv_dual: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_win_conv_dual: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE(ttsk_win_conv_supported(Cin, Cout, K), "ttsk_win_conv_dual: no instance for Cin=%d Cout=%d K=%d", Cin, Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_packed) | ((uintptr_t)bias) | ((uintptr_t)out_f32)) & 15) == 0 && (((uintptr_t)out_bf16) & 7) == 0,
               "ttsk_win_conv_dual: 16-byte alignment (bf16 copy: 8)");
  TTSK_REQUIRE((Cout & 3) == 0, "ttsk_win_conv_dual: Cout must be a multiple of 4");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_packed, bias, out_f32, S, K, Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr};
  a.out16 = (bf16_t*)out_bf16;
  launch_win_conv(a, B, S, Cin, 1, 1, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_hifi_upsample_win_supported(int Cin, int Cout, int stride) {
  // a wave's 32 (or 16) channels must lie in one phase, a workgroup's channel group on whole phases
  if (stride == 8) return (Cin == 256 || Cin == 512) && Cout > 0 && (Cout % WC_COUT == 0 || Cout == 128 || Cout == 64);
  if (stride == 2) return Cin == 128 && Cout == 64;
  return 0;
}
extern "C" int ttsk_hifi_upsample8_supported(int Cin, int Cout) { return ttsk_hifi_upsample_win_supported(Cin, Cout, 8); }

// ConvTranspose1d(Cin -> Cout, kernel 2 * stride, padding stride / 2) on 16-bit rows: see WcArgs::ups_cout.  w_packed:
// ttsk_win_conv_pack_* of the (stride * Cout, 2, Cin) tap-major pseudo-weight (tts_king_amd/ops.py:hifi_upsample_win_pack); bias_rep: the
// bias repeated for the `stride` phases.
extern "C" int ttsk_hifi_upsample_win(const void* x16, const void* w_packed, const float* bias_rep, void* out16, int f16, int B, int T, int Cin,
                                      int Cout, int stride, void* stream) {
  TTSK_REQUIRE(x16 && w_packed && bias_rep && out16, "ttsk_hifi_upsample_win: null pointer");
  TTSK_REQUIRE(B > 0 && T > 0 && B <= 65535, "ttsk_hifi_upsample_win: bad sizes B=%d T=%d", B, T);
  TTSK_REQUIRE(ttsk_hifi_upsample_win_supported(Cin, Cout, stride), "ttsk_hifi_upsample_win: no instance for Cin=%d Cout=%d stride=%d", Cin, Cout, stride);
  TTSK_REQUIRE(f16 == 1, "ttsk_hifi_upsample_win: built for fp16 rows (HiFi-GAN inference)");
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w_packed) | ((uintptr_t)bias_rep) | ((uintptr_t)out16)) & 15) == 0, "ttsk_hifi_upsample_win: 16-byte alignment");
  TTSK_REQUIRE((int64_t)B * T * stride * Cout * 2 < ((int64_t)1 << 40), "ttsk_hifi_upsample_win: sizes out of range");
  WcArgs a{(const bf16_t*)x16, (const bf16_t*)w_packed, bias_rep, out16, T, 2, stride * Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr, nullptr, nullptr,
           Cout, stride / 2};
  launch_win_conv(a, B, T, Cin, 0, 1, (hipStream_t)stream, 1);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
// HiFi-GAN's conv_pre, Conv1d(80 -> Cout, k) + LeakyReLU (hifi/models.py:152,186 and the first F.leaky_relu of :188) on fp16 rows, on the
// window-conv kernel's 96-channel instance (the contraction zero-padded from 80): x16 (B, T, 80) -> out16 (B, T, Cout) = lrelu(conv + bias).
// w_packed: ttsk_win_conv_pack_items of the (Cout, k, 80) tap-major weight.  Replaces a 3,072-row implicit GEMM (21 us of launch-bound work).
extern "C" int ttsk_hifi_conv_pre_win_supported(int Cin, int Cout, int K) {
  return Cin == 80 && Cout > 0 && Cout % WC_COUT == 0 && K >= 1 && K <= 2 * WC_H + 1 && (K & 1) == 1;
}
extern "C" int ttsk_hifi_conv_pre_win(const void* x16, const void* w_packed, const float* bias, void* out16, int f16, int B, int T, int Cin, int Cout,
                                      int K, float slope, void* stream) {
  TTSK_REQUIRE(x16 && w_packed && bias && out16, "ttsk_hifi_conv_pre_win: null pointer");
  TTSK_REQUIRE(B > 0 && T > 0 && B <= 65535, "ttsk_hifi_conv_pre_win: bad sizes B=%d T=%d", B, T);
  TTSK_REQUIRE(ttsk_hifi_conv_pre_win_supported(Cin, Cout, K), "ttsk_hifi_conv_pre_win: no instance for Cin=%d Cout=%d K=%d", Cin, Cout, K);
  TTSK_REQUIRE(f16 == 1, "ttsk_hifi_conv_pre_win: built for fp16 rows (HiFi-GAN inference)");
  TTSK_REQUIRE(slope >= 0.f && slope <= 1.f, "ttsk_hifi_conv_pre_win: slope in [0, 1] (0: no activation)");
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w_packed) | ((uintptr_t)bias) | ((uintptr_t)out16)) & 15) == 0, "ttsk_hifi_conv_pre_win: 16-byte alignment");
  WcArgs a{(const bf16_t*)x16, (const bf16_t*)w_packed, bias, out16, T, K, Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr};
  a.lrelu_slope = slope;
  launch_win_conv(a, B, T, Cin, 0, 1, (hipStream_t)stream, 1);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// The same operator, operands and pack for Cin = 256 on ups_loop_kernel: a workgroup per 96-frame tile loops over the stride * Cout / 256
// channel groups (HiFi-GAN's 256 -> 128 stride-8 upsampler: 49 -> see DESIGN.md 7.1).
extern "C" int ttsk_hifi_upsample_loop_supported(int Cin, int Cout, int stride) {
  return Cin == 256 && stride == 8 && Cout > 0 && Cout % 32 == 0;      // (8 * Cout is then a multiple of the 256-channel group; a wave's 32 channels lie in one phase)
}
extern "C" int ttsk_hifi_upsample_loop(const void* x16, const void* w_packed, const float* bias_rep, void* out16, int f16, int B, int T, int Cin,
                                       int Cout, int stride, void* stream) {
  TTSK_REQUIRE(x16 && w_packed && bias_rep && out16, "ttsk_hifi_upsample_loop: null pointer");
  TTSK_REQUIRE(B > 0 && T > 0 && (int64_t)B * ((T + 95) / 96) < ((int64_t)1 << 31), "ttsk_hifi_upsample_loop: bad sizes B=%d T=%d", B, T);
  TTSK_REQUIRE(ttsk_hifi_upsample_loop_supported(Cin, Cout, stride), "ttsk_hifi_upsample_loop: no instance for Cin=%d Cout=%d stride=%d", Cin, Cout, stride);
  TTSK_REQUIRE(f16 == 1, "ttsk_hifi_upsample_loop: built for fp16 rows (HiFi-GAN inference)");
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w_packed) | ((uintptr_t)bias_rep) | ((uintptr_t)out16)) & 15) == 0, "ttsk_hifi_upsample_loop: 16-byte alignment");
  TTSK_REQUIRE((int64_t)B * T * stride * Cout * 2 < ((int64_t)1 << 40), "ttsk_hifi_upsample_loop: sizes out of range");
  WcArgs a{(const bf16_t*)x16, (const bf16_t*)w_packed, bias_rep, out16, T, 2, stride * Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr, nullptr, nullptr,
           Cout, stride / 2};
  a.B = B;
  a.tiles_per_utt = (T + 95) / 96;
  hipLaunchKernelGGL((ups_loop_kernel<256, 96>), dim3(a.tiles_per_utt * B), dim3(WC_NT), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
extern "C" int ttsk_hifi_upsample8(const void* x16, const void* w_packed, const float* bias8, void* out16, int f16, int B, int T, int Cin,
                                   int Cout, void* stream) {
  return ttsk_hifi_upsample_win(x16, w_packed, bias8, out16, f16, B, T, Cin, Cout, 8, stream);
}

extern "C" int ttsk_win_conv_stats_rows(int B, int S) { return B * ((S + 63) / 64); }

// ttsk_win_conv with fp32 output whose BatchNorm statistics partials come out of the same kernel (Cin = 512)
extern "C" int ttsk_win_conv_stats(const void* x_bf16, const void* w_packed, const float* bias, float* out_f32, float* stats,
                                   const int32_t* frame_limit, int B, int S, int Cin, int Cout, int K, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_packed && out_f32 && stats, "ttsk_win_conv_stats: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_win_conv_stats: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE((Cin == 512 || Cin == 80) && ttsk_win_conv_supported(Cin, Cout, K), "ttsk_win_conv_stats: built for Cin = 512 or 80 (got Cin=%d Cout=%d K=%d)", Cin, Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_packed) | ((uintptr_t)bias) | ((uintptr_t)out_f32)) & 15) == 0, "ttsk_win_conv_stats: 16-byte alignment");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_packed, bias, out_f32, S, K, Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr, stats, frame_limit};
  launch_win_conv(a, B, S, Cin, 1, 1, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// ttsk_win_conv (Cin = 512, bf16 output, no gate / delta) that also emits the BatchNorm-backward statistics partials of the layer whose
// upstream gradient it produces: see WcArgs::bnb_x.
extern "C" int ttsk_win_conv_bnb(const void* x_bf16, const void* w_packed, void* out_bf16, float* stats, const float* bn_x_f32,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, const uint8_t* keep, float p,
                                 int use_tanh, const int32_t* frame_limit, int B, int S, int Cin, int Cout, int K, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_packed && out_bf16 && stats && bn_x_f32 && mean && rstd && gamma && beta, "ttsk_win_conv_bnb: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_win_conv_bnb: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE((Cin == 512 || Cin == 80) && Cout % WC_COUT == 0 && ttsk_win_conv_supported(Cin, Cout, K), "ttsk_win_conv_bnb: built for Cin = 512 or 80, Cout a multiple of 256 (got Cin=%d Cout=%d K=%d)", Cin, Cout, K);
  TTSK_REQUIRE(p >= 0.f && p < 1.f && (p == 0.f || keep), "ttsk_win_conv_bnb: dropout needs the forward's keep bits");
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_packed) | ((uintptr_t)out_bf16) | ((uintptr_t)bn_x_f32) | ((uintptr_t)mean) | ((uintptr_t)rstd) |
                 ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "ttsk_win_conv_bnb: 16-byte alignment");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_packed, nullptr, out_bf16, S, K, Cout, 0, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr, nullptr, frame_limit,
           0, 0, bn_x_f32, mean, rstd, gamma, beta, keep, stats, p, use_tanh};
  launch_win_conv(a, B, S, Cin, 0, 1, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_ffn_conv_fwd(const void* x_bf16, const void* w_bf16, const float* bias, void* out_bf16, int B, int S, int Cin, int Cout,
                                 int K, int relu, int packed, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_bf16 && bias && out_bf16, "ttsk_ffn_conv_fwd: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_ffn_conv_fwd: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE(ttsk_ffn_conv_supported(Cin, Cout, K), "ttsk_ffn_conv_fwd: no instance for Cin=%d Cout=%d K=%d", Cin, Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_bf16) | ((uintptr_t)bias) | ((uintptr_t)out_bf16)) & 15) == 0, "ttsk_ffn_conv_fwd: 16-byte alignment");
  TTSK_REQUIRE((int64_t)B * S * (Cout > Cin ? Cout : Cin) * 2 < ((int64_t)1 << 40), "ttsk_ffn_conv_fwd: sizes out of range");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_bf16, bias, out_bf16, S, K, Cout, relu, 0, 0, nullptr, 1, Cin, 0, nullptr, nullptr};
  launch_win_conv(a, B, S, Cin, 0, packed, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_win_conv_split(const void* x_bf16, const void* w_packed, float* slabs, int nsplit, int B, int S, int Cin_total, int Cout,
                                   int K, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_packed && slabs && nsplit >= 1 && nsplit <= 8, "ttsk_win_conv_split: bad arguments");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535 && Cin_total == nsplit * 256, "ttsk_win_conv_split: Cin must be nsplit * 256 (got %d, nsplit %d)", Cin_total, nsplit);
  TTSK_REQUIRE(ttsk_win_conv_supported(256, Cout, K), "ttsk_win_conv_split: no instance for Cout=%d K=%d", Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_packed) | ((uintptr_t)slabs)) & 15) == 0, "ttsk_win_conv_split: 16-byte alignment");
  WcArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_packed, nullptr, slabs, S, K, Cout, 0, 0, 0, nullptr, nsplit, Cin_total, (int64_t)B * S * Cout,
           nullptr, nullptr};
  launch_win_conv(a, B, S, 256, 1, 1, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
