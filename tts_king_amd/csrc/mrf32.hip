// mrf32.hip — HiFi-GAN's last stage in ONE kernel: the whole multi-receptive-field fusion at C = 32 (three ResBlock1s with kernel
// sizes 3 / 7 / 11 on the SAME input, summed, averaged), the LeakyReLU(0.01) that follows, conv_post (32 -> 1, k = 7) and tanh.
// reference: hifi/models.py:190-199 (Generator.forward: xs = sum_j resblocks[i*3+j](x); x = xs / 3; leaky_relu; conv_post; tanh),
// :88-95 (ResBlock1.forward).
//
// Why: at C = 32 a frame is 64 bytes and a conv tap 2 KFLOP per frame — the stage is 98,304 frames x 8 utterances = 50 MB per tensor
// pass, and as three resblock1_kernel launches + conv_post it made nine such passes (x in three times, the running sum out / in / out /
// in / out, the activated average in again: 432 + 55 MB by PMC against 50 MB in + 3 MB of waveform out).  Here a workgroup loads its x
// tile ONCE (386 output frames + 63 frames of halo per side: 60 for the k = 11 block, 3 for conv_post), keeps the raw tile in
// registers (fp16 pairs in MFMA accumulator layout), runs the three blocks one after the other on the two LDS windows of
// resblock.hip's kernel (XL = lrelu(x), TL = lrelu(t); residual in fp32 registers), adds each block's output to a register-resident
// sum with exactly the roundings the three-launch path makes (every block output and every partial sum is an fp16 tensor there), writes
// the activated average back into the XL window and runs conv_post + tanh from it: HBM sees x once and the waveform once.
//
// Weights: 18 convs, K taps x 2 KiB each, in resblock.hip's fragment-major packs, streamed through two LDS buffers (one for the c1
// convs, one for the c2 convs), each stage requested one conv pair ahead into its register set — across block boundaries too, so the
// k = 3 block's last pair already pulls the k = 7 block's first weights.
// Bit-identical to ttsk_hifi_resblock1 x 3 (modes 0 / 1 / 2) + ttsk_hifi_conv_post (tests/test_hifigan_gpu.py).
#include <type_traits>
#include "common.h"
#include "conv_post.h"

namespace {

constexpr int M_C = 32, M_NW = 8, M_NT = M_NW * 64, M_NTILE = 32, M_NSLOT = M_NTILE / M_NW, M_ROWS = M_NTILE * 16;
constexpr int M_G = 32;                          // guard rows either side (>= max tap reach 5 * 5 = 25)
constexpr int M_LROWS = M_ROWS + 2 * M_G;        // 576
constexpr int M_RS = M_C * 2;                    // 64-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 3 (resblock.hip)
constexpr int M_KP = 7, M_HP = 60 + (M_KP - 1) / 2;   // halo per side: 6 * (11 - 1) for the k = 11 block + conv_post's 3
constexpr int M_TT = M_ROWS - 2 * M_HP;          // 386 frames stored per workgroup
constexpr int M_NC = M_C / 16, M_CH8 = M_C / 8;
constexpr int M_TAP = M_NC * 1024;               // 2 KiB of weights per tap
constexpr int M_KMAX = 11, M_WSTAGE = M_KMAX * M_TAP;                // 22,528 B
constexpr int M_NLD = (M_WSTAGE + M_NT * 16 - 1) / (M_NT * 16);      // 3 x 16-byte loads per thread for the largest stage
constexpr int M_SMEM = 2 * M_LROWS * M_RS + 2 * M_WSTAGE + M_KP * M_C * 4;   // 119,680 B: one workgroup per CU

struct MrfArgs {
  const bf16_t* x;          // (B, len, 32) 16-bit: the raw stage input (the last upsampler's output)
  float* out;               // (B, 1, len) fp32: tanh(conv_post(lrelu(mean of the three blocks, 0.01)))
  bf16_t* stage_out;        // optional (B, len, 32) 16-bit: the activated average itself (tests); null in the product path
  const bf16_t* w[18];      // block j conv m at [6 j + m], m = c1_0, c2_0, c1_1, c2_1, c1_2, c2_2: fragment-major packs
  const float* b[18];
  const bf16_t* wpost;      // (1, 7, 32) tap-major 16-bit
  const float* bpost;
  int dil[3][3];
  int len;
  float slope, final_slope, scale;
};

template <int K0, int K1, int K2, bool F16>
__global__ __launch_bounds__(M_NT, 1) void mrf32_post_kernel(const MrfArgs a) {
  constexpr int C = M_C, NT = M_NT, NW = M_NW, NTILE = M_NTILE, NSLOT = M_NSLOT, G = M_G, LROWS = M_LROWS, RS = M_RS, NC = M_NC,
                CH8 = M_CH8, NLD = M_NLD, WSTAGE = M_WSTAGE, HP = M_HP, TT = M_TT, NCONV = 18;
  static_assert(K0 <= M_KMAX && K1 <= M_KMAX && K2 <= M_KMAX && 5 * ((M_KMAX - 1) / 2) <= M_G, "tap reach exceeds the guard rows");
  __shared__ __attribute__((aligned(16))) unsigned char smem[M_SMEM];
  unsigned char* XL = smem;
  unsigned char* TL = smem + LROWS * RS;
  unsigned char* WB = smem + 2 * LROWS * RS;
  float* wpf = (float*)(WB + 2 * WSTAGE);          // conv_post's weights as fp32 [7][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y, t0 = blockIdx.x * TT;
  const int len = a.len;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;
  const float slope = a.slope;
  auto swz = [](int row) __attribute__((always_inline)) { return (row >> 1) & 3; };

  // ---- weight stage g = conv g of the 18: K(g) taps of 2 KiB.  Two register sets: set A carries the c1 convs' stages (even g, LDS
  //      buffer 0), set B the c2 convs' (odd g, buffer 1); a stage is requested one conv pair before it is written to LDS.  Everything
  //      here is branch-free per lane — offsets are clamped, never predicated: a load inside a conditional makes hipcc drain the
  //      queue (s_waitcnt vmcnt(0)) where the branches join, which serialised every stage behind an L2 round trip.
  auto stage_bytes = [](int g) __attribute__((always_inline)) { return (g < 6 ? K0 : (g < 12 ? K1 : K2)) * M_TAP; };
  static_assert(NLD == 3, "three 16-byte loads per thread and stage");
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 struct is copied by memcpy, which kept
  struct Stage { u32x4 r0, r1, r2; };                                    //  these in scratch / LDS instead of registers)
  Stage wrA, wrB;
  auto load_stage = [&](int g, Stage& wr) __attribute__((always_inline)) {
    g = g < NCONV ? g : NCONV - 1;                 // (past the last conv: a harmless re-read, never stored)
    const int last16 = stage_bytes(g) - 16;
    const unsigned char* src = (const unsigned char*)a.w[g];
    const int o0 = tid * 16, o1 = (NT + tid) * 16, o2 = (2 * NT + tid) * 16;
    wr.r0 = *(const u32x4*)(src + (o0 < last16 ? o0 : last16));       // lanes past the stage re-read its last 16 bytes
    wr.r1 = *(const u32x4*)(src + (o1 < last16 ? o1 : last16));
    wr.r2 = *(const u32x4*)(src + (o2 < last16 ? o2 : last16));
  };
  auto store_stage = [&](unsigned char* dst, const Stage& wr) __attribute__((always_inline)) {
    // the whole buffer is written (lanes past a short stage store bytes nobody reads); only the lanes past the BUFFER are masked
    *(u32x4*)(dst + tid * 16) = wr.r0;
    *(u32x4*)(dst + (NT + tid) * 16) = wr.r1;
    if ((2 * NT + tid) * 16 < WSTAGE) *(u32x4*)(dst + (2 * NT + tid) * 16) = wr.r2;
  };
  unsigned char* WB0 = WB;
  unsigned char* WB1 = WB + WSTAGE;

  // ---- the x tile: rows r <-> frames t0 - 63 + r; TL <- raw x (read back below into registers), XL <- lrelu(x); zeros outside the
  //      utterance (the convs' zero padding) and in the guard rows.  The tile's loads go out FIRST (loads return in order and nothing
  //      starts before the tile is in LDS), the first weight stages behind them.
  {
    constexpr int NCH = (LROWS * CH8 + NT - 1) / NT;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int r = row - G;
      const int t = t0 - HP + r;
      const bool ok = idx < LROWS * CH8 && r >= 0 && r < NTILE * 16 && t >= 0 && t < len;
      const uint4 v = *(const uint4*)(xb + (ok ? (int64_t)t * C + ch * 8 : (int64_t)0));     // one select, no branch around the load
      xv[it] = ok ? v : make_uint4(0, 0, 0, 0);
    }
    load_stage(0, wrA);
    load_stage(1, wrB);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < LROWS * CH8) {
        const int pc = (ch ^ swz(row)) * 16;
        *(uint4*)(TL + row * RS + pc) = xv[it];
        *(uint4*)(XL + row * RS + pc) = lrelu8_fast<F16>(xv[it], slope);
      }
    }
  }
  for (int i = tid; i < M_KP * C; i += NT) wpf[i] = unpack1<F16>(a.wpost[i]);
  store_stage(WB0, wrA);
  load_stage(2, wrA);
  __syncthreads();

  // ---- this lane's elements of the tile — frame tile i = s * 8 + wave, frame i*16 + l15, channels c*16 + q*4 .. +3 (the MFMA
  //      accumulator layout) — as raw 16-bit pairs: every block starts from them
  uint2 x16[NC][NSLOT];
  const int own_off = (l15 + G) * RS + (q & 1) * 8;       // + i * 16 * RS + (((c*2 + (q>>1)) ^ swz(l15)) << 4)
  int own[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) own[c] = own_off + (((c * 2 + (q >> 1)) ^ swz(l15)) << 4) + wave * 16 * RS;       // + s * NW * 16 * RS
#pragma unroll
  for (int s = 0; s < NSLOT; ++s)
#pragma unroll
    for (int c = 0; c < NC; ++c) x16[c][s] = *(const uint2*)(TL + own[c] + s * NW * 16 * RS);
  __syncthreads();

  f32x4 xr[NC][NSLOT];         // the block's running x (fp32 residual)
  uint2 ysum[NC][NSLOT];       // the MRF sum so far, 16-bit pairs: what the three-launch path keeps in `out` between launches
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      float x0, x1, x2, x3;
      unpack2<F16>(x16[c][s].x, x0, x1); unpack2<F16>(x16[c][s].y, x2, x3);
      xr[c][s] = f32x4{x0, x1, x2, x3};
      ysum[c][s] = make_uint2(0u, 0u);
    }

  // One conv of a block over the whole tile: acc[c][s] += sum_tap W[tap] x in[row + (tap - HK) d].  Every slot is multiplied, also
  // tiles no valid output depends on (their rows read in-bounds guard rows): straight-line code lets the LDS reads run ahead.
  auto conv_taps = [&](auto Kc, const unsigned char* in, const unsigned char* wbuf, const int d, f32x4 (&acc)[NC][NSLOT]) __attribute__((always_inline)) {
    constexpr int K = decltype(Kc)::value, HK = (K - 1) / 2;
    const unsigned char* inl = in + (l15 + G) * RS + wave * 16 * RS;
    const unsigned char* wb = wbuf + lane * 16;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) acc[c][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < K; ++tap) {
      bf16x8 Af[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) Af[c] = *(const bf16x8*)(wb + (tap * NC + c) * 1024);
      const int shift = (tap - HK) * d;
      const unsigned char* inp = inl + shift * RS + ((q ^ swz(l15 + shift + 64)) << 4);
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const bf16x8 Bf = *(const bf16x8*)(inp + s * NW * 16 * RS);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c][s] = mfma16<F16>(Af[c], Bf, acc[c][s]);
      }
    }
  };
  auto frame_ok = [&](int s) __attribute__((always_inline)) {       // positions outside the utterance are conv zero padding
    const int t = t0 - HP + (s * NW + wave) * 16 + l15;
    return (t >= 0 && t < len) ? 1.f : 0.f;
  };

  // One ResBlock1 (three conv pairs) with kernel size K, as block number rb of the three; gs0 = its first conv's index of the 18.
  // `innerc`: the whole tile (halo included) lies inside the utterance, so no position is conv zero padding and the `* okf` of every epilogue is a
  // multiplication by one: left out (a fifth of the epilogues' VALU instructions, and at C = 32 the epilogues issue as many cycles as the MFMAs)
  auto block = [&](auto Kc, const int rb, auto innerc) __attribute__((always_inline)) {
    constexpr int K = decltype(Kc)::value, HK = (K - 1) / 2;
    constexpr bool INNER = decltype(innerc)::value;
    const int gs0 = 6 * rb;
    // the block's outputs are needed on rows [60, 452) (conv_post's reach); conv j's on that range widened by what the later convs
    // consume: rows [S, ROWS - S) after `S += HK * d`.  The k = 11 block starts at 0: it needs the whole tile.
    int S = (HP - (M_KP - 1) / 2) - 6 * (K - 1);
#pragma unroll 1
    for (int pair = 0; pair < 3; ++pair) {
      const int gs = gs0 + 2 * pair;
      f32x4 acc[NC][NSLOT];
      // ---- c1: dilated conv of XL = lrelu(x) -> TL = lrelu(. + b)
      {
        const int d = a.dil[rb][pair];
        S += HK * d;
        const int tlo = S >> 4, thi = (M_ROWS - S + 15) >> 4;
        f32x4 bv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) bv[c] = *(const f32x4*)(a.b[gs] + c * 16 + q * 4);
        conv_taps(Kc, XL, WB0, d, acc);
        store_stage(WB1, wrB);                    // c2's weights (requested a pair ago) into the buffer the previous c2 read
        load_stage(gs + 3, wrB);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          const int i = s * NW + wave;
          if (i >= tlo && i < thi) {
            const float okf = frame_ok(s);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              f32x4 v = acc[c][s] + bv[c];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = INNER ? fmaxf(v[e], v[e] * slope) : fmaxf(v[e], v[e] * slope) * okf;
              *(uint2*)(TL + own[c] + s * NW * 16 * RS) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
            }
          }
        }
        __syncthreads();
      }
      // ---- c2: conv (dilation 1) of TL, + b + x -> the new x; XL = lrelu(x) for the next pair
      {
        S += HK;
        const int tlo = S >> 4, thi = (M_ROWS - S + 15) >> 4;
        f32x4 bv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) bv[c] = *(const f32x4*)(a.b[gs + 1] + c * 16 + q * 4);
        conv_taps(Kc, TL, WB1, 1, acc);
        store_stage(WB0, wrA);                    // the next c1's weights
        load_stage(gs + 4, wrA);
        if (pair < 2) {
#pragma unroll
          for (int s = 0; s < NSLOT; ++s) {
            const int i = s * NW + wave;
            if (i >= tlo && i < thi) {
              const float okf = frame_ok(s);
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                f32x4 v = acc[c][s] + bv[c] + xr[c][s];
                if (!INNER) v = v * okf;
                xr[c][s] = v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
                *(uint2*)(XL + own[c] + s * NW * 16 * RS) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
              }
            }
          }
        } else {
          // the block's output y = c2_2(..) + x, rounded to 16 bits as the three-launch path stores it, joins the MRF sum in this
          // lane's registers: block 0: sum = y; block 1: sum = fp16(sum + y); block 2: sum = fp16(lrelu((sum + y) * scale, final_slope))
#pragma unroll
          for (int s = 0; s < NSLOT; ++s) {
            const float okf = frame_ok(s);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              f32x4 v = acc[c][s] + bv[c] + xr[c][s];
              if (!INNER) v = v * okf;
              uint2 y = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
              if (rb > 0) {
                const unsigned yw[2] = {y.x, y.y}, ow[2] = {ysum[c][s].x, ysum[c][s].y};
                unsigned rw[2];
                const float sc = rb == 2 ? a.scale : 1.f, fsl = rb == 2 ? a.final_slope : 1.f;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  float vl, vh, ol, oh;
                  unpack2<F16>(yw[e], vl, vh); unpack2<F16>(ow[e], ol, oh);
                  vl = (vl + ol) * sc; vh = (vh + oh) * sc;
                  vl = vl > 0.f ? vl : vl * fsl; vh = vh > 0.f ? vh : vh * fsl;
                  rw[e] = pack2<F16>(vl, vh);
                }
                y = make_uint2(rw[0], rw[1]);
              }
              ysum[c][s] = y;
              if (rb < 2) {
                // the next block starts from the raw tile again: XL <- lrelu(x) on this lane's elements (XL was last read by this pair's c1)
                float x0, x1, x2, x3;
                unpack2<F16>(x16[c][s].x, x0, x1); unpack2<F16>(x16[c][s].y, x2, x3);
                xr[c][s] = f32x4{x0, x1, x2, x3};
                *(uint2*)(XL + own[c] + s * NW * 16 * RS) = make_uint2(lrelu2_fast<F16>(x16[c][s].x, slope), lrelu2_fast<F16>(x16[c][s].y, slope));
              } else {
                *(uint2*)(XL + own[c] + s * NW * 16 * RS) = y;          // the activated MRF average: conv_post's input
              }
            }
          }
        }
        __syncthreads();
      }
    }
  };
  if (t0 - HP >= 0 && t0 - HP + M_ROWS <= len) {
    block(std::integral_constant<int, K0>{}, 0, std::true_type{});
    block(std::integral_constant<int, K1>{}, 1, std::true_type{});
    block(std::integral_constant<int, K2>{}, 2, std::true_type{});
  } else {
    block(std::integral_constant<int, K0>{}, 0, std::false_type{});
    block(std::integral_constant<int, K1>{}, 1, std::false_type{});
    block(std::integral_constant<int, K2>{}, 2, std::false_type{});
  }

  // ---- XL rows [60, 452) hold lrelu(mean, 0.01) for frames t0 - 3 .. t0 + 389 (zero outside the utterance): conv_post + tanh, one
  //      output sample per thread, in ttsk_hifi_conv_post's summation order
  if (a.stage_out) {
    bf16_t* __restrict__ ob = a.stage_out + (int64_t)bi * len * C;
    for (int idx = tid; idx < TT * CH8; idx += NT) {
      const int rr = idx / CH8, ch = idx - rr * CH8;
      const int t = t0 + rr, row = rr + HP + G;
      if (t < len) *(uint4*)(ob + (int64_t)t * C + ch * 8) = *(const uint4*)(XL + row * RS + ((ch ^ swz(row)) << 4));
    }
  }
  if (tid < TT && t0 + tid < len) {
    float accp = a.bpost[0];
#pragma unroll
    for (int j = 0; j < M_KP; ++j) {
      const int row = tid + HP - (M_KP - 1) / 2 + j + G;
      const unsigned char* rowp = XL + row * RS;
      const float* wj = wpf + j * C;
      const int sw = swz(row);
#pragma unroll
      for (int ch = 0; ch < CH8; ++ch) {
        const uint4 v = *(const uint4*)(rowp + ((ch ^ sw) << 4));
        accp += conv_post_dot8<F16>(v, wj + ch * 8);
      }
    }
    a.out[(int64_t)bi * len + t0 + tid] = tanhf(accp);
  }
}

}  // namespace

extern "C" int ttsk_hifi_mrf32_post_supported(int C, int k0, int k1, int k2, int k_post) {
  return C == 32 && k0 == 3 && k1 == 7 && k2 == 11 && k_post == M_KP;
}

extern "C" int ttsk_hifi_mrf32_post(const void* x16, float* out, void* stage_out16, int f16, const void* const* weights /* 18 packs */,
                                    const float* const* biases /* 18 x [32] */, const int32_t* dilations /* 3 x 3 */,
                                    const void* w_post16, const float* b_post, int B, int len, int C, int k0, int k1, int k2, int k_post,
                                    float slope, float final_slope, float scale, void* stream) {
  TTSK_REQUIRE(x16 && out && weights && biases && dilations && w_post16 && b_post, "ttsk_hifi_mrf32_post: null pointer");
  TTSK_REQUIRE(ttsk_hifi_mrf32_post_supported(C, k0, k1, k2, k_post),
               "ttsk_hifi_mrf32_post: built for C = 32, resblock kernel sizes (3, 7, 11), conv_post k = 7 (got C=%d, k=(%d,%d,%d), post %d)", C, k0, k1,
               k2, k_post);
  TTSK_REQUIRE(B > 0 && len > 0 && B <= 65535, "ttsk_hifi_mrf32_post: bad sizes B=%d len=%d", B, len);
  TTSK_REQUIRE(slope > 0.f && slope < 1.f, "ttsk_hifi_mrf32_post: LeakyReLU slope %g outside (0, 1)", slope);
  MrfArgs a;
  a.x = (const bf16_t*)x16; a.out = out; a.stage_out = (bf16_t*)stage_out16;
  for (int i = 0; i < 18; ++i) {
    TTSK_REQUIRE(weights[i] && biases[i], "ttsk_hifi_mrf32_post: null weight/bias %d", i);
    TTSK_REQUIRE((((uintptr_t)weights[i]) & 15) == 0 && (((uintptr_t)biases[i]) & 15) == 0, "ttsk_hifi_mrf32_post: 16-byte alignment");
    a.w[i] = (const bf16_t*)weights[i];
    a.b[i] = biases[i];
  }
  TTSK_REQUIRE((((uintptr_t)x16) & 15) == 0 && (!stage_out16 || (((uintptr_t)stage_out16) & 15) == 0), "ttsk_hifi_mrf32_post: 16-byte alignment");
  for (int j = 0; j < 3; ++j) {
    const int32_t* d = dilations + 3 * j;
    TTSK_REQUIRE(d[0] >= 1 && d[1] >= 1 && d[2] >= 1 && d[0] + d[1] + d[2] <= 9 && d[0] <= 5 && d[1] <= 5 && d[2] <= 5,
                 "ttsk_hifi_mrf32_post: dilations (%d,%d,%d) of block %d exceed the tile halo built for (1,3,5)", d[0], d[1], d[2], j);
    for (int m = 0; m < 3; ++m) a.dil[j][m] = d[m];
  }
  a.wpost = (const bf16_t*)w_post16; a.bpost = b_post;
  a.len = len; a.slope = slope; a.final_slope = final_slope; a.scale = scale;
  dim3 grid((len + M_TT - 1) / M_TT, B);
  if (f16) hipLaunchKernelGGL((mrf32_post_kernel<3, 7, 11, true>), grid, dim3(M_NT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((mrf32_post_kernel<3, 7, 11, false>), grid, dim3(M_NT), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
