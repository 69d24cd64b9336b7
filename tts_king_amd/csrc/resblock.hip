// resblock.hip — HiFi-GAN ResBlock1, all six convolutions of the block fused into one kernel per time tile.
// reference: hifi/models.py:12-95 (ResBlock1.forward :88-95), called from Generator.forward :190-196.
//
//   for m in 0..2:   t = conv_{k, dil d_m}(lrelu(x)) ;  x = conv_{k, dil 1}(lrelu(t)) + x
//
// One 256-thread workgroup owns TT output frames of one utterance plus the halo the six convolutions consume
// (H = (k-1)/2 * (d0+d1+d2+3) frames per side).  The tile lives in LDS for the whole block: XL = lrelu(x) and
// TL = lrelu(t) as bf16 [frame][channel] rows (padded by 32 B: conflict-free ds_read_b128 fragments), the residual
// x itself stays in fp32 REGISTERS in MFMA accumulator layout, so HBM sees one read of x and one write of the result
// per ResBlock instead of 5 tensor passes per conv pair.
//
// Each conv is D[cout][frame] = sum_tap sum_cin W[cout][tap][cin] * IN[frame + (tap-(k-1)/2)*d][cin] on
// v_mfma_f32_16x16x32_bf16: the weight fragment (A) comes straight from global/L2 into registers one tap ahead, the
// activation fragment (B) is one ds_read_b128 of 16 frames x 8 channels shifted by the tap, and is reused for every
// 16-channel output tile.  Frames on the MFMA column axis make every lane own 4 consecutive output channels of one
// frame: epilogues write 8-byte channel runs.  A wave owns frame tiles (wave, wave+4, ...) for ALL convs, so the
// residual of an element is always in the lane that produces its update.
#include "common.h"

namespace {

struct RbArgs {
  const bf16_t* x;
  bf16_t* out;
  const bf16_t* w[6];   // convs1[0], convs2[0], convs1[1], convs2[1], convs1[2], convs2[2]: (C, K, C) bf16
  const float* b[6];
  int dil[3];
  int len;              // frames per utterance
  int mode;             // 0: out = y   1: out += y   2: out = (out + y) * scale
  float scale;
  float slope;
};

template <int C, int K, int TT>
struct RbGeom {
  static constexpr int HK = (K - 1) / 2;
  static constexpr int H = 6 * (K - 1);               // halo per side for dilations (1,3,5): HK * (1+3+5+3)
  static constexpr int ROWS = TT + 2 * H;
  static constexpr int NTILE = (ROWS + 15) / 16;
  static constexpr int NSLOT = (NTILE + 3) / 4;
  static constexpr int G = 32;                         // guard rows either side (>= max tap reach 5*HK = 25)
  static constexpr int LROWS = NTILE * 16 + 2 * G;
  static constexpr int RS = C * 2 + 32;                // padded row stride in bytes
  static constexpr int NC = C / 16;
  static constexpr int KS = C / 32;
  static constexpr int SMEM = 2 * LROWS * RS;
};

template <int C, int K, int TT, bool F16>
__global__ __launch_bounds__(256, 1) void resblock1_kernel(const RbArgs a) {
  using Gm = RbGeom<C, K, TT>;
  constexpr int HK = Gm::HK, H = Gm::H, NTILE = Gm::NTILE, NSLOT = Gm::NSLOT, G = Gm::G, LROWS = Gm::LROWS, RS = Gm::RS,
                NC = Gm::NC, KS = Gm::KS, CH8 = C / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Gm::SMEM];
  unsigned char* XL = smem;
  unsigned char* TL = smem + LROWS * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int len = a.len;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;
  const float slope = a.slope;

  // ---- stage the tile: TL <- raw x (only to initialise the residual registers), XL <- lrelu(x); zeros outside
  //      the utterance (conv zero padding) and in the guard rows
  for (int idx = tid; idx < LROWS * CH8; idx += 256) {
    const int row = idx / CH8, ch = idx - row * CH8;
    const int r = row - G;
    const int t = t0 - H + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r >= 0 && r < NTILE * 16 && t >= 0 && t < len) v = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    *(uint4*)(TL + row * RS + ch * 16) = v;
    *(uint4*)(XL + row * RS + ch * 16) = lrelu8<F16>(v, slope);
  }
  __syncthreads();

  // ---- residual registers: xr[c][s] = x[frame = tile(s)*16 + l15][channels c*16 + q*4 .. +3]
  f32x4 xr[NC][NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int i = s * 4 + wave;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint2 v = make_uint2(0, 0);
      if (i < NTILE) v = *(const uint2*)(TL + (i * 16 + l15 + G) * RS + (c * 16 + q * 4) * 2);
      float x0, x1, x2, x3;
      unpack2<F16>(v.x, x0, x1); unpack2<F16>(v.y, x2, x3);
      xr[c][s] = f32x4{x0, x1, x2, x3};
    }
  }
  __syncthreads();

  int S = 0;   // halo consumed so far: conv outputs are needed (and valid) on rows [S, ROWS - S)
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int d = half == 0 ? a.dil[m] : 1;
      S += HK * d;
      const int tlo = S >> 4, thi = (Gm::ROWS - S + 15) >> 4;
      const unsigned char* in = half == 0 ? XL : TL;
      const bf16_t* __restrict__ w = a.w[m * 2 + half];
      const float* __restrict__ bias = a.b[m * 2 + half];

      f32x4 acc[NC][NSLOT];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) acc[c][s] = f32x4{0.f, 0.f, 0.f, 0.f};

      // weight fragments for one tap: lane holds W[cout = c*16 + l15][tap][cin = ks*32 + q*8 .. +7]
      const bf16_t* wl = w + (int64_t)l15 * K * C + q * 8;
      bf16x8 Af[2][NC * KS];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Af[0][c * KS + ks] = *(const bf16x8*)(wl + (int64_t)c * 16 * K * C + ks * 32);

      const unsigned char* inl = in + (l15 + G) * RS + q * 16;
#pragma unroll
      for (int tap = 0; tap < K; ++tap) {
        if (tap + 1 < K) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
              Af[(tap + 1) & 1][c * KS + ks] = *(const bf16x8*)(wl + (int64_t)c * 16 * K * C + (tap + 1) * C + ks * 32);
        }
        const int shift = (tap - HK) * d;
        const unsigned char* inp = inl + shift * RS;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          const int i = s * 4 + wave;
          if (i >= tlo && i < thi) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
              for (int c = 0; c < NC; ++c)
                acc[c][s] = mfma16<F16>(Af[tap & 1][c * KS + ks], Bf, acc[c][s]);
            }
          }
        }
      }

      // ---- epilogue
      const bool last = (m == 2 && half == 1);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 bv = *(const f32x4*)(bias + c * 16 + q * 4);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          const int i = s * 4 + wave;
          if (i >= tlo && i < thi) {
            const int t = t0 - H + i * 16 + l15;
            const bool ok = t >= 0 && t < len;        // positions outside the utterance are conv zero padding
            f32x4 v = acc[c][s] + bv;
            unsigned char* dst;
            if (half == 0) {
              dst = TL;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = ok ? (v[e] > 0.f ? v[e] : v[e] * slope) : 0.f;
            } else {
              dst = XL;
              v += xr[c][s];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
              xr[c][s] = v;
              if (!last) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
              }
            }
            *(uint2*)(dst + (i * 16 + l15 + G) * RS + (c * 16 + q * 4) * 2) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- XL rows [H, H+TT) now hold the block output (bf16): coalesced 16-byte copy-out
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
  for (int idx = tid; idx < TT * CH8; idx += 256) {
    const int rr = idx / CH8, ch = idx - rr * CH8;
    const int t = t0 + rr;
    if (t >= len) break;
    uint4 v = *(const uint4*)(XL + (rr + H + G) * RS + ch * 16);
    uint4* op = (uint4*)(ob + (int64_t)t * C + ch * 8);
    if (a.mode != 0) {
      const uint4 o = *op;
      const unsigned vw[4] = {v.x, v.y, v.z, v.w}, ow[4] = {o.x, o.y, o.z, o.w};
      unsigned rw[4];
      const float sc = a.mode == 2 ? a.scale : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float vl, vh, ol, oh;
        unpack2<F16>(vw[e], vl, vh); unpack2<F16>(ow[e], ol, oh);
        rw[e] = pack2<F16>((vl + ol) * sc, (vh + oh) * sc);
      }
      v = make_uint4(rw[0], rw[1], rw[2], rw[3]);
    }
    *op = v;
  }
}

template <int C, int K, int TT>
int launch_rb(const RbArgs& a, int B, int f16, hipStream_t s) {
  dim3 grid((a.len + TT - 1) / TT, B);
  if (f16) hipLaunchKernelGGL((resblock1_kernel<C, K, TT, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((resblock1_kernel<C, K, TT, false>), grid, dim3(256), 0, s, a);
  return 0;
}

}  // namespace

extern "C" int ttsk_hifi_resblock1(const void* x_bf16, void* out_bf16, int f16, const void* const* weights /* 6 x (C,K,C) */,
                                   const float* const* biases /* 6 x [C] */, const int32_t* dilations /* 3 */, int B, int len,
                                   int C, int K, int mode, float scale, float slope, void* stream) {
  TTSK_REQUIRE(x_bf16 && out_bf16 && weights && biases && dilations, "ttsk_hifi_resblock1: null pointer");
  TTSK_REQUIRE(B > 0 && len > 0 && B <= 65535, "ttsk_hifi_resblock1: bad sizes B=%d len=%d", B, len);
  TTSK_REQUIRE(mode >= 0 && mode <= 2, "ttsk_hifi_resblock1: mode");
  TTSK_REQUIRE(dilations[0] >= 1 && dilations[1] >= 1 && dilations[2] >= 1 && dilations[0] + dilations[1] + dilations[2] <= 9 &&
                   dilations[0] <= 5 && dilations[1] <= 5 && dilations[2] <= 5,
               "ttsk_hifi_resblock1: dilations (%d,%d,%d) exceed the tile halo built for (1,3,5)", dilations[0], dilations[1], dilations[2]);
  RbArgs a;
  a.x = (const bf16_t*)x_bf16;
  a.out = (bf16_t*)out_bf16;
  for (int i = 0; i < 6; ++i) {
    TTSK_REQUIRE(weights[i] && biases[i], "ttsk_hifi_resblock1: null weight/bias %d", i);
    TTSK_REQUIRE((((uintptr_t)weights[i]) & 15) == 0 && (((uintptr_t)biases[i]) & 15) == 0, "ttsk_hifi_resblock1: 16-byte alignment");
    a.w[i] = (const bf16_t*)weights[i];
    a.b[i] = biases[i];
  }
  TTSK_REQUIRE((((uintptr_t)x_bf16) & 15) == 0 && (((uintptr_t)out_bf16) & 15) == 0, "ttsk_hifi_resblock1: 16-byte alignment");
  for (int i = 0; i < 3; ++i) a.dil[i] = dilations[i];
  a.len = len; a.mode = mode; a.scale = scale; a.slope = slope;
  hipStream_t s = (hipStream_t)stream;
  const int key = C * 100 + K;
  switch (key) {
    case 3203: launch_rb<32, 3, 256>(a, B, f16, s); break;
    case 3207: launch_rb<32, 7, 256>(a, B, f16, s); break;
    case 3211: launch_rb<32, 11, 256>(a, B, f16, s); break;
    case 6403: launch_rb<64, 3, 128>(a, B, f16, s); break;
    case 6407: launch_rb<64, 7, 128>(a, B, f16, s); break;
    case 6411: launch_rb<64, 11, 128>(a, B, f16, s); break;
    default:
      ttsk_set_error("ttsk_hifi_resblock1: no fused instance for C=%d K=%d (C in {32,64}, K in {3,7,11})", C, K);
      return TTSK_EINVAL;
  }
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_hifi_resblock1_supported(int C, int K) {
  return (C == 32 || C == 64) && (K == 3 || K == 7 || K == 11);
}
