// convwin.hip — "window" Conv1d for the 128-channel HiFi-GAN stage: one kernel per convolution, the activation window
// of a time tile resident in LDS, the taps as shifted LDS reads, the weights streamed.
// reference: hifi/models.py:88-95 (the convs of ResBlock1 at the C = 128 stage, where six convs no longer fit one tile).
//
// Why not the implicit-GEMM kernel: there a K step re-fetches the same activation rows from L2 for every tap (a
// 256x128x64 step moves 48 KiB per 4.2 MFLOP and both tile configurations are bound by the CU's operand fetch rate,
// 25-28 % MFMA utilisation).  Here a workgroup loads TT + 2*32 frames x 128 channels ONCE (92 KiB with padding), and per
// tap only the 32 KiB of weights (fragment-major pack, coalesced, 4 stages ahead in registers, double-buffered in LDS):
// 4x fewer fetched bytes per FLOP.  D[cout][frame] orientation, fragment layouts, padding and the weight pipeline are
// those of resblock.hip.
#include "common.h"

namespace {

struct CwArgs {
  const bf16_t* x;      // (B, len, C) 16-bit, already activated (conv zero padding = zeros outside [0, len))
  const bf16_t* w;      // fragment-major pack [K][C/32][C/16][64][8]
  const float* bias;
  const bf16_t* R;      // residual (B, len, C) or null
  bf16_t* out;          // (B, len, C)
  bf16_t* out2;         // optional second output = lrelu(out, slope)
  int len, K, dil;
  int lrelu_out;        // out = lrelu(v, slope)
  float slope;
};

constexpr int CW_C = 128, CW_TT = 256, CW_H = 32, CW_NW = 8, CW_NT = CW_NW * 64;
constexpr int CW_LROWS = CW_TT + 2 * CW_H;          // 320
constexpr int CW_RS = CW_C * 2 + 32;                // 288 B per row
constexpr int CW_ACT = CW_LROWS * CW_RS;            // 92,160 B
constexpr int CW_NC = CW_C / 16, CW_KS = CW_C / 32; // 8 cout tiles, 4 k-steps
constexpr int CW_WTAP = CW_NC * CW_KS * 1024;       // 32 KiB per tap
constexpr int CW_NLD = CW_WTAP / (CW_NT * 16);      // 4 x 16-byte loads per thread per tap
constexpr int CW_SMEM = CW_ACT + 2 * CW_WTAP;       // 157,696 B
constexpr int CW_NSLOT = (CW_TT / 16) / CW_NW;      // 2 frame tiles per wave

template <bool F16>
__global__ __launch_bounds__(CW_NT, 1) void conv_window_kernel(const CwArgs a) {
  constexpr int C = CW_C, TT = CW_TT, H = CW_H, RS = CW_RS, NC = CW_NC, KS = CW_KS, NT = CW_NT, NLD = CW_NLD, CH8 = C / 8,
                NSLOT = CW_NSLOT, NW = CW_NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[CW_SMEM];
  unsigned char* XW = smem;
  unsigned char* WB = smem + CW_ACT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y, t0 = blockIdx.x * TT;
  const int len = a.len, K = a.K, d = a.dil;
  const int HK = (K - 1) / 2;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;

  // ---- weight pipeline: tap g travels in register set g % 4, requested four taps ahead, stored to LDS one tap ahead
  uint4 wr0[NLD], wr1[NLD], wr2[NLD], wr3[NLD];
  auto load_tap = [&](int g, uint4 (&wr)[NLD]) __attribute__((always_inline)) {
    const unsigned char* src = (const unsigned char*)a.w + (int64_t)g * CW_WTAP;
#pragma unroll
    for (int i = 0; i < NLD; ++i) wr[i] = *(const uint4*)(src + (i * NT + tid) * 16);
  };
  auto store_tap = [&](int g, uint4 (&wr)[NLD]) __attribute__((always_inline)) {
    unsigned char* dst = WB + (g & 1) * CW_WTAP;
#pragma unroll
    for (int i = 0; i < NLD; ++i) *(uint4*)(dst + (i * NT + tid) * 16) = wr[i];
  };
#pragma unroll
  for (int i = 0; i < NLD; ++i) wr0[i] = wr1[i] = wr2[i] = wr3[i] = make_uint4(0, 0, 0, 0);
  load_tap(0, wr0);
  if (K > 1) load_tap(1, wr1);
  if (K > 2) load_tap(2, wr2);
  if (K > 3) load_tap(3, wr3);

  // ---- activation window, all loads in flight together
  {
    constexpr int NCH = (CW_LROWS * CH8 + NT - 1) / NT;     // 10
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - H + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < CW_LROWS * CH8 && t >= 0 && t < len) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < CW_LROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = xv[it];
    }
  }
  store_tap(0, wr0);
  if (K > 4) load_tap(4, wr0);
  // bias and residual are requested now and consumed in the epilogue
  f32x4 bv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) bv[c] = *(const f32x4*)(a.bias + c * 16 + q * 4);
  __syncthreads();

  f32x4 acc[NC][NSLOT];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[c][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* inl = XW + (l15 + H) * RS + q * 16;
#pragma unroll 1
  for (int g = 0; g < K; ++g) {
    const unsigned char* wb = WB + (g & 1) * CW_WTAP + lane * 16;
    const unsigned char* inp = inl + (g - HK) * d * RS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 Af[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) Af[c] = *(const bf16x8*)(wb + (ks * NC + c) * 1024);
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const int i = s * NW + wave;
        const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c][s] = mfma16<F16>(Af[c], Bf, acc[c][s]);
      }
    }
    switch ((g + 1) & 3) {
      case 0: if (g + 1 < K) store_tap(g + 1, wr0); if (g + 5 < K) load_tap(g + 5, wr0); break;
      case 1: if (g + 1 < K) store_tap(g + 1, wr1); if (g + 5 < K) load_tap(g + 5, wr1); break;
      case 2: if (g + 1 < K) store_tap(g + 1, wr2); if (g + 5 < K) load_tap(g + 5, wr2); break;
      default: if (g + 1 < K) store_tap(g + 1, wr3); if (g + 5 < K) load_tap(g + 5, wr3); break;
    }
    __syncthreads();
  }

  // ---- epilogue: bias, residual, activation in registers (fp32), then staged through the window buffer (its rows are
  //      dead now) for full-row 16-byte stores
  const bf16_t* __restrict__ Rb = a.R ? a.R + (int64_t)bi * len * C : nullptr;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int i = s * NW + wave;
    const int t = t0 + i * 16 + l15;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 v = acc[c][s] + bv[c];
      if (Rb && t < len) {
        const uint2 r = *(const uint2*)(Rb + (int64_t)t * C + c * 16 + q * 4);
        float r0, r1, r2, r3;
        unpack2<F16>(r.x, r0, r1); unpack2<F16>(r.y, r2, r3);
        v += f32x4{r0, r1, r2, r3};
      }
      if (a.lrelu_out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
      }
      *(uint2*)(XW + (i * 16 + l15) * RS + (c * 16 + q * 4) * 2) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
    }
  }
  __syncthreads();
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
  bf16_t* __restrict__ ob2 = a.out2 ? a.out2 + (int64_t)bi * len * C : nullptr;
  constexpr int NCO = TT * CH8 / NT;     // 8
#pragma unroll
  for (int it = 0; it < NCO; ++it) {
    const int idx = it * NT + tid;
    const int rr = idx / CH8, ch = idx - rr * CH8;
    const int t = t0 + rr;
    if (t >= len) continue;
    const uint4 v = *(const uint4*)(XW + rr * RS + ch * 16);
    *(uint4*)(ob + (int64_t)t * C + ch * 8) = v;
    if (ob2) *(uint4*)(ob2 + (int64_t)t * C + ch * 8) = lrelu8<F16>(v, a.slope);
  }
}

}  // namespace

extern "C" int ttsk_hifi_conv_window_supported(int C, int K, int dil) {
  return C == CW_C && K >= 1 && K <= 11 && (K & 1) == 1 && dil >= 1 && dil * ((K - 1) / 2) <= CW_H;
}

extern "C" int ttsk_hifi_conv_window(const void* x16, const void* w_pack, const float* bias, const void* R16, void* out16,
                                     void* out2_16, int f16, int B, int len, int C, int K, int dil, int lrelu_out, float slope,
                                     void* stream) {
  TTSK_REQUIRE(x16 && w_pack && bias && out16, "ttsk_hifi_conv_window: null pointer");
  TTSK_REQUIRE(B > 0 && len > 0 && B <= 65535, "ttsk_hifi_conv_window: bad sizes");
  TTSK_REQUIRE(ttsk_hifi_conv_window_supported(C, K, dil), "ttsk_hifi_conv_window: no instance for C=%d K=%d dil=%d", C, K, dil);
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w_pack) | ((uintptr_t)bias) | ((uintptr_t)out16) | ((uintptr_t)R16) | ((uintptr_t)out2_16)) & 15) == 0,
               "ttsk_hifi_conv_window: 16-byte alignment");
  CwArgs a{(const bf16_t*)x16, (const bf16_t*)w_pack, bias, (const bf16_t*)R16, (bf16_t*)out16, (bf16_t*)out2_16, len, K, dil, lrelu_out, slope};
  dim3 grid((len + CW_TT - 1) / CW_TT, B);
  if (f16) hipLaunchKernelGGL(conv_window_kernel<true>, grid, dim3(CW_NT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv_window_kernel<false>, grid, dim3(CW_NT), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
