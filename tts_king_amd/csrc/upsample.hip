// upsample.hip — the stride-2 ConvTranspose1d upsamplers of the HiFi-GAN generator (kernel 4, padding 1: 128 -> 64 and
// 64 -> 32 channels) as one streaming kernel.  reference: hifi/models.py:166-176 (self.ups), :189 (x = self.ups[i](x)).
//
// On the implicit-GEMM kernel each of the two output phases is a launch of its own: the input is read twice, the 64- or
// 32-channel output fills half or a quarter of a 128-column tile, and the rows of one phase are written with stride 2
// (79 + 115 us for the two stages against ~20 us each of HBM traffic).  Here a workgroup loads 256 + 2 input frames ONCE,
// keeps all four taps of the weights in LDS as MFMA A-fragments, computes both phases
//     out[2t]   = x[t] W[1] + x[t-1] W[3]          out[2t+1] = x[t+1] W[0] + x[t] W[2]
// with frames on the MFMA columns (D[cout][frame], as in resblock.hip) and writes its 512 output frames as one contiguous
// block through LDS.  HBM sees one read of the input and one write of the output.
#include "common.h"

namespace {

struct UpArgs {
  const bf16_t* x;      // (B, T, CIN) 16-bit, already activated by its producer
  const bf16_t* w;      // (4, COUT, CIN) tap-major (ttsk_pack_conv_weight mode 1)
  const float* bias;    // [COUT]
  bf16_t* out;          // (B, 2T, COUT)
  int T;
};

template <int CIN, int COUT, bool F16>
__global__ __launch_bounds__(256) void ups2_kernel(const UpArgs a) {
  constexpr int TT = 256;                       // input frames per workgroup
  constexpr int XR = TT + 2;                    // + one halo frame per side
  constexpr int XRS = CIN * 2 + 32;             // row stride = 2 mod 4 sixteen-byte units: the 16 lanes of a ds_read_b128 group (8 rows at k-chunk q, the 8 OTHER
                                                // rows at q + 1) cover all 16 units; + 16 (an odd stride, round 2) collided two ways on every fragment read: 49 % of this
                                                // kernel's LDS cycles at 64 channels (profiles/r05_mfma_util.json)
  constexpr int KS = CIN / 32, NC = COUT / 16;
  constexpr int WBYTES = 4 * KS * NC * 1024;    // all taps as A fragments [tap][k-step][cout tile][lane][8]
  constexpr int ORS = COUT * 2 + 16;            // output staging row stride
  constexpr int XBYTES = XR * XRS, OBYTES = 2 * TT * ORS;
  constexpr int ABYTES = XBYTES > OBYTES ? XBYTES : OBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[ABYTES + WBYTES];
  unsigned char* XW = smem;                     // input window, later the output tile
  unsigned char* WB = smem + ABYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * TT, T = a.T;
  const bf16_t* __restrict__ xb = a.x + (int64_t)b * T * CIN;

  // ---- every global load of the tile in flight before the first LDS store
  constexpr int CH8 = CIN / 8;
  constexpr int NX = (XR * CH8 + 255) / 256, NWL = WBYTES / (256 * 16);
  uint4 xv[NX], wv[NWL];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = i * 256 + tid, row = idx / CH8, ch = idx - row * CH8;
    const int t = t0 - 1 + row;
    xv[i] = make_uint4(0, 0, 0, 0);
    if (idx < XR * CH8 && t >= 0 && t < T) xv[i] = *(const uint4*)(xb + (int64_t)t * CIN + ch * 8);
  }
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    // fragment f = ((tap * KS + ks) * NC + c), lane l: W[tap][co][ks*32 + (l >> 4)*8 .. +8] with the tile's rows PERMUTED: row m = l & 15 of tile c is
    // output channel co = NC*4 * (m >> 2) + 4 c + (m & 3), so that a lane's NC accumulator tiles (rows 4 lg .. 4 lg + 3 each) are NC*4 CONSECUTIVE
    // channels of one frame: 16-byte staging stores, conflict-free at the 16-byte-multiple row stride (8-byte columns 160 bytes apart collided four ways)
    const int e = i * 256 + tid, f = e >> 6, l = e & 63;
    const int c = f % NC, ks = (f / NC) % KS, tap = f / (NC * KS);
    const int co = NC * 4 * ((l & 15) >> 2) + 4 * c + (l & 3);
    wv[i] = *(const uint4*)(a.w + ((int64_t)(tap * COUT + co) * CIN + ks * 32 + (l >> 4) * 8));
  }
  float bv[NC][4];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[c][r] = a.bias[NC * 4 * lg + 4 * c + r];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = i * 256 + tid, row = idx / CH8, ch = idx - row * CH8;
    if (idx < XR * CH8) *(uint4*)(XW + row * XRS + ch * 16) = xv[i];
  }
#pragma unroll
  for (int i = 0; i < NWL; ++i) *(uint4*)(WB + (i * 256 + tid) * 16) = wv[i];
  __syncthreads();

  // ---- both phases of two frame tiles at a time (the A fragments of a tap are shared by the two tiles)
  // phase 1 (odd outputs): taps 0 (x[t+1]) and 2 (x[t]);  phase 0 (even outputs): taps 1 (x[t]) and 3 (x[t-1])
  constexpr int NFT = TT / 16;
  unsigned packed[NFT / 8][2][2][NC][2];   // [tile pair][tile][phase][cout tile][2 dwords = 4 channels]
#pragma unroll
  for (int tp = 0; tp < NFT / 8; ++tp) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      f32x4 acc[2][NC];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int tap = ph == 1 ? 2 * q : 2 * q + 1;
        const int sh = ph == 1 ? 1 - q : -q;                 // input frame offset of this tap
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 Af[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) Af[c] = *(const bf16x8*)(WB + ((tap * KS + ks) * NC + c) * 1024 + lane * 16);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ft = (tp * 2 + u) * 4 + wave;          // frame tile of this wave
            const bf16x8 Bf = *(const bf16x8*)(XW + (ft * 16 + l15 + 1 + sh) * XRS + (ks * 32 + lg * 8) * 2);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = mfma16<F16>(Af[c], Bf, acc[u][c]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          packed[tp][u][ph][c][0] = pack2<F16>(acc[u][c][0] + bv[c][0], acc[u][c][1] + bv[c][1]);
          packed[tp][u][ph][c][1] = pack2<F16>(acc[u][c][2] + bv[c][2], acc[u][c][3] + bv[c][3]);
        }
    }
  }
  __syncthreads();                               // every wave is done reading the input window: it becomes the output tile
#pragma unroll
  for (int tp = 0; tp < NFT / 8; ++tp)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int c = 0; c < NC; c += 2) {        // two tiles = 8 consecutive channels = one 16-byte store
          const int ft = (tp * 2 + u) * 4 + wave;
          const int orow = 2 * (ft * 16 + l15) + ph;
          *(uint4*)(XW + orow * ORS + (NC * 4 * lg + 4 * c) * 2) =
              make_uint4(packed[tp][u][ph][c][0], packed[tp][u][ph][c][1], packed[tp][u][ph][c + 1][0], packed[tp][u][ph][c + 1][1]);
        }
  __syncthreads();
  // ---- 2*TT output frames of COUT channels: one contiguous block of the (B, 2T, COUT) tensor
  constexpr int OCH8 = COUT / 8;
  bf16_t* __restrict__ ob = a.out + ((int64_t)b * 2 * T + 2 * t0) * COUT;
  const int nrows = (2 * (T - t0)) < 2 * TT ? 2 * (T - t0) : 2 * TT;
  if constexpr (OCH8 == 4) {
    // A wave copies 16 rows x 4 chunks = 1 KiB of the output per pass.  Which lane takes which (row, chunk) is free (the global store is the same
    // contiguous KiB), so it follows the LDS: a ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
    // same + 32 — and at an 80-byte row stride 16 DIFFERENT rows of one chunk column cover all 16 sixteen-byte units (5 is odd), while the row-major
    // assignment put rows r, r + 3, r + 5, r + 6 in a group: two-way conflicts on every read.
    const int l = lane & 31;
    const int grp = ((l >= 4 && l < 12) || (l >= 16 && l < 20) || l >= 28) ? 1 : 0;
    const int j = grp ? (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16)) : (l < 4 ? l : (l < 16 ? l - 8 : l - 12));
    const int ch = grp + 2 * (lane >> 5);
    for (int r0 = wave * 16; r0 < nrows; r0 += 64) {
      const int row = r0 + j;
      if (row < nrows) *(uint4*)(ob + (int64_t)row * COUT + ch * 8) = *(const uint4*)(XW + row * ORS + ch * 16);
    }
  } else {
    for (int idx = tid; idx < nrows * OCH8; idx += 256) {
      const int row = idx / OCH8, ch = idx - row * OCH8;
      *(uint4*)(ob + (int64_t)row * COUT + ch * 8) = *(const uint4*)(XW + row * ORS + ch * 16);
    }
  }
}

template <int CIN, int COUT>
void launch_ups2(const UpArgs& a, int B, int f16, hipStream_t s) {
  dim3 grid((a.T + 255) / 256, B);
  if (f16) hipLaunchKernelGGL((ups2_kernel<CIN, COUT, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((ups2_kernel<CIN, COUT, false>), grid, dim3(256), 0, s, a);
}

}  // namespace

extern "C" int ttsk_hifi_upsample2_supported(int Cin, int Cout, int stride, int k) {
  return stride == 2 && k == 4 && ((Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 32)) ? 1 : 0;
}

extern "C" int ttsk_hifi_upsample2(const void* x16, const void* w16, const float* bias, void* out16, int f16, int B, int T,
                                   int Cin, int Cout, void* stream) {
  TTSK_REQUIRE(x16 && w16 && bias && out16 && B > 0 && B <= 65535 && T > 0, "ttsk_hifi_upsample2: bad arguments");
  TTSK_REQUIRE(ttsk_hifi_upsample2_supported(Cin, Cout, 2, 4), "ttsk_hifi_upsample2: instances are 128 -> 64 and 64 -> 32 channels (got %d -> %d)", Cin, Cout);
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w16) | ((uintptr_t)out16)) & 15) == 0, "ttsk_hifi_upsample2: 16-byte alignment");
  UpArgs a{(const bf16_t*)x16, (const bf16_t*)w16, bias, (bf16_t*)out16, T};
  if (Cin == 128) launch_ups2<128, 64>(a, B, f16, (hipStream_t)stream);
  else launch_ups2<64, 32>(a, B, f16, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
