// ffn_conv.hip — the FFT block's first position-wise conv, forward: h = relu(Conv1d(256 -> d_ff, k = 9)(x)).
// reference: fs_two/transformer/SubLayers.py:93-101 (PositionwiseFeedForward.forward: w_1 on x^T, relu), called once per
// FFTBlock (Layers.py:25-34); 62.6 % of the step's FLOPs sit in this conv and its two gradients.
//
// The implicit-GEMM kernels re-fetch the activation rows of a tile for every tap and synchronise their eight waves per K step
// (0.9-1.3 us per step against 0.55 us of MFMA time; 45 us for the decoder's 31.9 GFLOP).  This kernel is the HiFi-GAN window
// kernel's layout (convwin.hip) for this shape: a workgroup owns 112 frames of one utterance x 256 of the output channels; the
// 120-row activation window (all 256 input channels, 65 KiB) is loaded into LDS once; each of the eight waves owns 32 output
// channels and streams its weights L2 -> registers, one (tap, 128-channel half) step = 8 fragments = 32 VGPRs, requested two steps ahead (three register sets);
// no barrier inside the tap loop.  T = 423: 4 tiles x 16 utterances x 4 channel groups = 256 workgroups, one per CU.
// The weights are read from the tap-major bf16 shadow (Cout, k, 256) as it is: a fragment is 16 rows x 64 contiguous bytes.
#include "common.h"

namespace {

constexpr int FC_CIN = 256, FC_TT = 112, FC_H = 4, FC_NW = 8, FC_NT = FC_NW * 64;
constexpr int FC_RS = FC_CIN * 2 + 32;                 // 544 B per window row
constexpr int FC_XROWS = FC_TT + 2 * FC_H;             // 120
constexpr int FC_SMEM = FC_XROWS * FC_RS;              // 65,280 B
constexpr int FC_NF = FC_TT / 16, FC_KH = 4, FC_CT = 2, FC_COUT = FC_NW * FC_CT * 16;   // 7 frame tiles; 256 output channels per workgroup

struct FfnArgs {
  const bf16_t* x;      // [B*S][256] bf16 (PAD rows are zeros)
  const bf16_t* w;      // (Cout, K, 256) bf16, tap-major
  const float* bias;    // [Cout]
  bf16_t* out;          // [B*S][Cout]
  int S, K, Cout, relu;
  int B, tiles_per_utt;
};

// (Cout, K, 256) tap-major -> [K][8][Cout/16][64][8] fragment-major, up to 16 weights of one shape per launch (blockIdx.y)
struct PackBatch {
  const bf16_t* src[16];
  bf16_t* dst[16];
};
__global__ __launch_bounds__(256) void ffn_pack_kernel(const PackBatch pb, int Cout, int K) {
  const bf16_t* __restrict__ src = pb.src[blockIdx.y];
  bf16_t* __restrict__ dst = pb.dst[blockIdx.y];
  const int64_t n8 = (int64_t)Cout * K * (FC_CIN / 8);           // 16-byte pieces
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const int l = (int)(i & 63);
    int64_t f = i >> 6;
    const int c = (int)(f % (Cout / 16)); f /= (Cout / 16);
    const int ks = (int)(f % (FC_CIN / 32));
    const int tap = (int)(f / (FC_CIN / 32));
    const int co = c * 16 + (l & 15), ci = ks * 32 + (l >> 4) * 8;
    *(uint4*)(dst + i * 8) = *(const uint4*)(src + ((int64_t)co * K + tap) * FC_CIN + ci);
  }
}

template <bool PACKED>
__global__ __launch_bounds__(FC_NT, 1) void ffn_conv_fwd_kernel(const FfnArgs a) {
  constexpr int C = FC_CIN, TT = FC_TT, RS = FC_RS, NT = FC_NT, CH8 = C / 8, KH = FC_KH, CT = FC_CT, NF = FC_NF;
  __shared__ __attribute__((aligned(16))) unsigned char XW[FC_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  // XCD-aware mapping: consecutive workgroup ids go round the 8 XCDs, and a channel group's weights (1.2 MB at d_ff = 1024, k = 9) should
  // stay in ONE XCD's 4 MiB L2 instead of all four groups (4.7 MB) in every L2: channel group = f(id % 8), tile = the rest.
  int bi, t0, cg;
  {
    const int id = blockIdx.x, ncg = a.Cout / FC_COUT, ntile = a.tiles_per_utt * a.B;
    const int xcd = id & 7, per = 8 / (ncg < 8 ? ncg : 8);        // XCDs per channel group (ncg = 1, 2, 4, 8); other counts: plain order
    int tile;
    if ((8 % (ncg < 8 ? ncg : 8)) == 0 && ncg <= 8 && (ntile * ncg) % 8 == 0 && ntile % per == 0) {
      cg = xcd / per;
      tile = (id >> 3) * per + (xcd % per);
    } else {
      cg = id / ntile;
      tile = id - cg * ntile;
    }
    bi = tile / a.tiles_per_utt;
    t0 = (tile - bi * a.tiles_per_utt) * TT;
  }
  const int S = a.S, K = a.K, HK = (K - 1) / 2, K2 = 2 * K;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * S * C;

  // step g of the 2K-step sequence: tap g / 2, input-channel half g & 1
  // tap-major weights (Cout, K, 256): a fragment is 16 rows x 64 B, 4,608 B apart (half of every 128-B line fetched is used);
  // PACKED: fragment-major [tap][k-step][cout tile][lane][8] (ttsk_ffn_pack_weight): a fragment is 1 KiB contiguous
  const bf16_t* wrow[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
    wrow[cc] = PACKED ? a.w + ((int64_t)(cg * (FC_COUT / 16) + wave * CT + cc) * 64 + lane) * 8
                      : a.w + ((int64_t)(cg * FC_COUT + (wave * CT + cc) * 16 + l15) * K) * C + q * 8;
  const int64_t kstep_stride = (int64_t)(a.Cout / 16) * 512;     // PACKED: elements per (tap, k-step)
  bf16x8 wa[KH][CT], wb[KH][CT], wc[KH][CT];      // three register sets: a step's weights are requested two steps (>= 1 us) ahead
  auto load_w = [&](int g, bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
    if (PACKED) {
      const int64_t off = (int64_t)((g >> 1) * (C / 32) + (g & 1) * KH) * kstep_stride;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) w[ks][cc] = *(const bf16x8*)(wrow[cc] + off + ks * kstep_stride);
    } else {
      const int off = (g >> 1) * C + (g & 1) * (KH * 32);
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) w[ks][cc] = *(const bf16x8*)(wrow[cc] + off + ks * 32);
    }
  };
  load_w(0, wa);
  load_w(1, wb);
  if (2 < K2) load_w(2, wc);

  {  // ---- activation window: rows t0 - 4 .. t0 + 116 of the utterance, zeros outside it (the conv's zero padding)
    constexpr int NCH = (FC_XROWS * CH8 + NT - 1) / NT;     // 8
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - FC_H + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < FC_XROWS * CH8 && t >= 0 && t < S) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < FC_XROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = xv[it];
    }
  }
  f32x4 bv[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) bv[cc] = *(const f32x4*)(a.bias + cg * FC_COUT + (wave * CT + cc) * 16 + q * 4);
  __syncthreads();

  f32x4 acc[CT][NF];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const unsigned char* inl = XW + (l15 + FC_H) * RS + q * 16;
    auto step = [&](int g, const bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
      const unsigned char* inp = inl + ((g >> 1) - HK) * RS + (g & 1) * (KH * 64);
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
          for (int cc = 0; cc < CT; ++cc) acc[cc][i] = mfma16<false>(w[ks][cc], Bf, acc[cc][i]);
        }
      }
    };
#pragma unroll 1
    for (int g = 0; g < K2; g += 3) {
      step(g, wa);
      if (g + 3 < K2) load_w(g + 3, wa);
      if (g + 1 < K2) {
        step(g + 1, wb);
        if (g + 4 < K2) load_w(g + 4, wb);
      }
      if (g + 2 < K2) {
        step(g + 2, wc);
        if (g + 5 < K2) load_w(g + 5, wc);
      }
    }
  }
  __syncthreads();          // every wave is done with the window: its rows become the output staging tile

#pragma unroll
  for (int i = 0; i < NF; ++i) {
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      f32x4 v = acc[cc][i] + bv[cc];
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      *(uint2*)(XW + (i * 16 + l15) * RS + ((wave * CT + cc) * 16 + q * 4) * 2) = make_uint2(pack2<false>(v[0], v[1]), pack2<false>(v[2], v[3]));
    }
  }
  __syncthreads();
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * S * a.Cout + cg * FC_COUT;
  constexpr int OCH = FC_COUT / 8;                       // 32 16-byte chunks per output row of this channel group
  constexpr int NCO = TT * OCH / NT;                     // 7
#pragma unroll
  for (int it = 0; it < NCO; ++it) {
    const int idx = it * NT + tid;
    const int rr = idx / OCH, ch = idx - rr * OCH;
    const int t = t0 + rr;
    if (t < S) *(uint4*)(ob + (int64_t)t * a.Cout + ch * 8) = *(const uint4*)(XW + rr * RS + ch * 16);
  }
}

}  // namespace

extern "C" int ttsk_ffn_conv_supported(int Cin, int Cout, int K) {
  return Cin == FC_CIN && Cout > 0 && Cout % FC_COUT == 0 && K >= 1 && K <= 2 * FC_H + 1 && (K & 1) == 1;
}

extern "C" int ttsk_ffn_pack_weight_batch(const void* const* w_bf16, void* const* packed_bf16, int n, int Cout, int K, void* stream) {
  TTSK_REQUIRE(w_bf16 && packed_bf16 && n > 0 && n <= 16 && Cout > 0 && Cout % 16 == 0 && K >= 1, "ttsk_ffn_pack_weight_batch: bad arguments");
  PackBatch pb;
  for (int i = 0; i < 16; ++i) {
    pb.src[i] = (const bf16_t*)w_bf16[i < n ? i : 0];
    pb.dst[i] = (bf16_t*)packed_bf16[i < n ? i : 0];
    TTSK_REQUIRE(pb.src[i] && pb.dst[i] && ((((uintptr_t)pb.src[i]) | ((uintptr_t)pb.dst[i])) & 15) == 0, "ttsk_ffn_pack_weight_batch: null / unaligned pointer");
  }
  const int64_t n8 = (int64_t)Cout * K * (FC_CIN / 8);
  int blocks = (int)((n8 + 255) / 256);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, pb, Cout, K);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_ffn_pack_weight(const void* w_bf16, void* packed_bf16, int Cout, int K, void* stream) {
  return ttsk_ffn_pack_weight_batch(&w_bf16, &packed_bf16, 1, Cout, K, stream);
}

extern "C" int ttsk_ffn_conv_fwd(const void* x_bf16, const void* w_bf16, const float* bias, void* out_bf16, int B, int S, int Cin, int Cout,
                                 int K, int relu, int packed, void* stream) {
  TTSK_REQUIRE(x_bf16 && w_bf16 && bias && out_bf16, "ttsk_ffn_conv_fwd: null pointer");
  TTSK_REQUIRE(B > 0 && S > 0 && B <= 65535, "ttsk_ffn_conv_fwd: bad sizes B=%d S=%d", B, S);
  TTSK_REQUIRE(ttsk_ffn_conv_supported(Cin, Cout, K), "ttsk_ffn_conv_fwd: no instance for Cin=%d Cout=%d K=%d", Cin, Cout, K);
  TTSK_REQUIRE(((((uintptr_t)x_bf16) | ((uintptr_t)w_bf16) | ((uintptr_t)bias) | ((uintptr_t)out_bf16)) & 15) == 0, "ttsk_ffn_conv_fwd: 16-byte alignment");
  TTSK_REQUIRE((int64_t)B * S * (Cout > Cin ? Cout : Cin) * 2 < ((int64_t)1 << 40), "ttsk_ffn_conv_fwd: sizes out of range");
  const int tpu = (S + FC_TT - 1) / FC_TT;
  FfnArgs a{(const bf16_t*)x_bf16, (const bf16_t*)w_bf16, bias, (bf16_t*)out_bf16, S, K, Cout, relu, B, tpu};
  dim3 grid(tpu * B * (Cout / FC_COUT));
  if (packed) hipLaunchKernelGGL(ffn_conv_fwd_kernel<true>, grid, dim3(FC_NT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(ffn_conv_fwd_kernel<false>, grid, dim3(FC_NT), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
