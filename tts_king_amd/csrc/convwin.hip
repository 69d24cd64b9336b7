// convwin.hip — "window" convolutions of the HiFi-GAN ResBlocks: the activation window of a time tile resident in LDS, the taps as
// shifted LDS reads, the weights streamed from L2 into registers.
// reference: hifi/models.py:88-95 (ResBlock1).  Kernels: one conv per launch at C = 128 (conv_window2_kernel), and the
// (c1 -> LeakyReLU -> c2 -> + x) pair per launch at C = 128 and C = 64 / 32 (conv_pair_kernel, conv_pair_fs_kernel).
//
// Why not the implicit-GEMM kernel: there a K step re-fetches the same activation rows from L2 for every tap (a
// 256x128x64 step moves 48 KiB per 4.2 MFLOP and both tile configurations are bound by the CU's operand fetch rate,
// 25-28 % MFMA utilisation).  Here a workgroup loads its frames x channels ONCE and per tap only the weights (fragment-major
// pack, coalesced).  D[cout][frame] orientation, fragment layouts and padding are those of resblock.hip.
#include <type_traits>
#include "common.h"
#include "tapring.h"
#include <cstdlib>

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

constexpr int CW_C = 128;
constexpr int CW_RS = CW_C * 2 + 32;                // 288 B per row
constexpr int CW_NC = CW_C / 16, CW_KS = CW_C / 32; // 8 cout tiles, 4 k-steps
constexpr int CW_WTAP = CW_NC * CW_KS * 1024;       // 32 KiB per tap

// ---- one conv per launch: 192-frame tiles, four waves, two workgroups per CU.
// (Round 1's kernel kept the weights in LDS: every wave read all 32 KiB of a tap, eight waves met at a barrier per tap, and with
// 157 KiB of LDS a CU held one workgroup: 24 us per launch + 5.5 us per tap.)  Here a wave owns 32 output channels and all 192
// frames of the tile: its weights (8 KiB per tap, fragment-major, so one coalesced 16-byte load per lane and fragment) go straight
// from L2 to registers two taps ahead, the tap loop has no barrier, LDS holds the window alone (244 rows, 70 KiB) and a second
// workgroup on the CU computes while this one loads or stores: 23 us per launch + 4.0 us per tap.  Tile lengths of 96 / 128
// frames at three workgroups per CU measured the same.  LDS reads per MFMA: 0.5 (12 activation fragments per 24 MFMAs).
constexpr int C2_H = 26, C2_NW = 4, C2_NT = C2_NW * 64;
constexpr int C2_CT = CW_NC / C2_NW;                // 2 cout tiles per wave

template <bool F16, int TT, int OCC>
__global__ __launch_bounds__(C2_NT, OCC) void conv_window2_kernel(const CwArgs a) {
  constexpr int C2_LROWS = TT + 2 * C2_H;           // 244 rows at TT = 192
  constexpr int C2_SMEM = C2_LROWS * CW_RS;         // 70,272 B
  constexpr int C = CW_C, H = C2_H, RS = CW_RS, NC = CW_NC, KS = CW_KS, NT = C2_NT, CH8 = C / 8, NF = TT / 16, CT = C2_CT;
  __shared__ __attribute__((aligned(16))) unsigned char XW[C2_SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y, t0 = blockIdx.x * TT;
  const int len = a.len, K = a.K, d = a.dil;
  const int HK = (K - 1) / 2;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;

  // ---- this wave's weights of one tap: fragments (ks, cout tile wave*CT + cc)
  bf16x8 wa[KS][CT], wb[KS][CT];
  const unsigned char* wsrc = (const unsigned char*)a.w + (wave * CT) * 1024 + lane * 16;
  auto load_w = [&](int g, bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {
    const unsigned char* src = wsrc + (int64_t)g * CW_WTAP;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) w[ks][cc] = *(const bf16x8*)(src + (ks * NC + cc) * 1024);
  };

  // ---- activation window, all loads in flight together
  {
    constexpr int NCH = (C2_LROWS * CH8 + NT - 1) / NT;     // 16
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - H + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < C2_LROWS * CH8 && t >= 0 && t < len) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
    // the weight fragments are requested BEHIND the window: loads return in order, and nothing starts before the window is in LDS
    load_w(0, wa);
    if (K > 1) load_w(1, wb);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < C2_LROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = xv[it];
    }
  }
  f32x4 bv[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) bv[cc] = *(const f32x4*)(a.bias + (wave * CT + cc) * 16 + q * 4);
  __syncthreads();

  f32x4 acc[CT][NF];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* inl = XW + (l15 + H) * RS + q * 16;
  auto tap = [&](int g, const bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {
    const unsigned char* inp = inl + (g - HK) * d * RS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) acc[cc][i] = mfma16<F16>(w[ks][cc], Bf, acc[cc][i]);
      }
    }
  };
#pragma unroll 1
  for (int g = 0; g < K; g += 2) {
    tap(g, wa);
    if (g + 2 < K) load_w(g + 2, wa);
    if (g + 1 < K) {
      tap(g + 1, wb);
      if (g + 3 < K) load_w(g + 3, wb);
    }
  }
  __syncthreads();          // every wave is done with the window: its rows become the output staging tile

  const bf16_t* __restrict__ Rb = a.R ? a.R + (int64_t)bi * len * C : nullptr;
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int t = t0 + i * 16 + l15;
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      const int co = (wave * CT + cc) * 16 + q * 4;
      f32x4 v = acc[cc][i] + bv[cc];
      if (Rb && t < len) {
        const uint2 r = *(const uint2*)(Rb + (int64_t)t * C + co);
        float r0, r1, r2, r3;
        unpack2<F16>(r.x, r0, r1); unpack2<F16>(r.y, r2, r3);
        v += f32x4{r0, r1, r2, r3};
      }
      if (a.lrelu_out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
      }
      *(uint2*)(XW + (i * 16 + l15) * RS + co * 2) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
    }
  }
  __syncthreads();
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
  bf16_t* __restrict__ ob2 = a.out2 ? a.out2 + (int64_t)bi * len * C : nullptr;
  constexpr int NCO = TT * CH8 / NT;     // 12
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

// ---- the (dilated conv -> LeakyReLU -> conv -> + x) pair of ResBlock1 in ONE kernel (hifi/models.py:88-95: xt = c1(lrelu(x));
// xt = c2(lrelu(xt)); x = xt + x).  Measured on the single-conv kernels above: a launch takes (time of its HBM traffic) +
// (time of its MFMAs) with no overlap between the two however the workgroups are sized or de-phased, and the pair moves
// 300 MB (x_l in, t out; t in, x in, x' out, lrelu(x') out) for 100 MB of necessary traffic.  Here a workgroup keeps t on
// chip: it loads x once for 96 + 2*33 frames (LeakyReLU applied on the way into LDS), computes c1 for 112 frames (the 96 of
// the tile + the halo c2 needs: 7 frame tiles instead of 6, 8 % more MFMAs over the pair), writes lrelu(t) as fp16 rows of
// a second LDS window (zero outside [0, len): c2's zero padding), runs c2 from there and adds the raw x it re-reads
// from L2 in the epilogue.  Same fp16 roundings and the same accumulation order as the two-launch path: bit-identical.
// Wave = 32 output channels x all frames, weights of both convs streamed L2 -> registers as one sequence of 2K taps.
constexpr int CP_TT = 96, CP_TH = 8, CP_XH = CP_TH + 25;            // t-window starts 8 frames before the tile, x-window 33
constexpr int CP_TROWS = CP_TT + 2 * CP_TH;                         // 112 = 7 frame tiles of c1 output
constexpr int CP_XROWS = CP_TROWS + 50;                             // 162
constexpr int CP_SMEM = (CP_XROWS + CP_TROWS) * CW_RS;              // 78,912 B: two workgroups per CU

struct PairArgs {
  const bf16_t* x;      // (B, len, C) 16-bit, raw block input
  const bf16_t* w1;     // fragment-major packs [K][C/32][C/16][64][8]
  const bf16_t* w2;
  const float* b1;
  const float* b2;
  bf16_t* out;          // (B, len, C): y = c2(lrelu(c1(lrelu(x)))) + x, stored per `mode`
  int len, K, dil;
  float slope;
  // MRF average folded into the last pair of each ResBlock (hifi/models.py:190-197: xs = sum_j resblock_j(x); x = xs / 3, then the
  // consumer's leaky_relu), with resblock.hip's modes:  0: out = y   1: out += y   2: out = lrelu((out + y) * scale, final_slope)
  int mode;
  float scale, final_slope;
#ifdef TTSK_STAMPS
  unsigned long long* stamps;   // diagnostic build only (make stamps; ttsk_hifi_conv_pair_set_stamps): 8 x s_memrealtime per workgroup
#endif
};
#ifdef TTSK_STAMPS
#define TTSK_STAMP(i)                                                                                            \
  do {                                                                                                            \
    if (a.stamps && threadIdx.x == 0)                                                                             \
      a.stamps[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#else
#define TTSK_STAMP(i) do {} while (0)      // the product library carries no stamp code and no global state for it
#endif

template <bool F16>
__global__ __launch_bounds__(C2_NT, 2) void conv_pair_kernel(const PairArgs a) {
  constexpr int C = CW_C, TT = CP_TT, RS = CW_RS, NC = CW_NC, KS = CW_KS, NT = C2_NT, CH8 = C / 8, CT = C2_CT;
  constexpr int NF1 = CP_TROWS / 16, NF2 = TT / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[CP_SMEM];
  unsigned char* XW = smem;
  unsigned char* TW = smem + CP_XROWS * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y, t0 = blockIdx.x * TT;
  const int len = a.len, K = a.K, d = a.dil;
  const int HK = (K - 1) / 2;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;

  // The wave's 32 output channels, rows of the A operand PERMUTED (round 6, as csrc/pairws.hip): row l15 of tile cc is channel
  // 32 * wave + 8 * (l15 >> 2) + 4 * cc + (l15 & 3), so that a lane's 4 + 4 accumulator rows of the two tiles are 8 CONSECUTIVE channels of
  // one frame — one 16-byte LDS / global store per frame tile and lane instead of two 8-byte ones (whose rows collided four ways in LDS:
  // 17 % of the kernel's LDS cycles), and the output needs no staging tile, barrier and copy-out.  Same products, same order: bit-identical.
  bf16x8 wa[KS][CT], wb[KS][CT];
  int woff[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int co = 32 * wave + 8 * (l15 >> 2) + 4 * cc + (l15 & 3);
    woff[cc] = (co >> 4) * 1024 + ((co & 15) + 16 * q) * 16;
  }
  const int co8 = 32 * wave + 8 * q;          // this lane's 8 output channels
  auto load_w = [&](int g, bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {      // tap g of the 2K-tap sequence c1 | c2
    const unsigned char* src = (const unsigned char*)(g < K ? a.w1 : a.w2) + (int64_t)(g < K ? g : g - K) * CW_WTAP;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) w[ks][cc] = *(const bf16x8*)(src + ks * NC * 1024 + woff[cc]);
  };
  TTSK_STAMP(0);

  {  // ---- x window: lrelu(x) rows t0 - 33 .. t0 + 129, zeros outside the utterance.  The window is sized for dilation 5 (c1 reaches 25
     // rows beyond the 112 it computes); a smaller dilation reads HK * d rows either side only — the others are never fetched (they
     // stay zero, nobody reads them): 122 / 142 / 162 rows for d = 1 / 3 / 5 at k = 11.
    constexpr int NCH = (CP_XROWS * CH8 + NT - 1) / NT;     // 11
    const int xlo = CP_XH - CP_TH - HK * d, xhi = CP_XH - CP_TH + CP_TROWS + HK * d;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - CP_XH + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < CP_XROWS * CH8 && t >= 0 && t < len && row >= xlo && row < xhi) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
    // the weight fragments are requested BEHIND the window: loads return in order, and nothing starts before the window is in LDS
    load_w(0, wa);
    load_w(1, wb);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < CP_XROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = lrelu8_fast<F16>(xv[it], a.slope);
    }
  }
  f32x4 bv1[CT], bv2[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    bv1[cc] = *(const f32x4*)(a.b1 + co8 + 4 * cc);
    bv2[cc] = *(const f32x4*)(a.b2 + co8 + 4 * cc);
  }
  TTSK_STAMP(1);
  __syncthreads();
  TTSK_STAMP(2);

  // ---- c1 over the 112 frames of the t window
  // (round 6: the accumulators are not zeroed — 104 x 4 v_mov per wave and workgroup, and every VALU instruction costs a quarter of an MFMA here,
  // DESIGN.md 8.2 — the first tap's first k-step multiplies onto the constant 0 instead: same sums, bit for bit)
  f32x4 acc[CT][NF1];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  {
    const unsigned char* inl = XW + (l15 + CP_XH - CP_TH) * RS + q * 16;      // t-window row r <-> x-window row r + 25
    auto tap1 = [&](auto firstc, int g, const bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {
      const unsigned char* inp = inl + (g - HK) * d * RS;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < NF1; ++i) {
          const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
          for (int cc = 0; cc < CT; ++cc) acc[cc][i] = mfma16<F16>(w[ks][cc], Bf, (decltype(firstc)::value && ks == 0) ? zero4 : acc[cc][i]);
        }
      }
    };
    // K is odd and >= 3: taps 0 .. K-2 in pairs (the first pair peeled: its first tap starts the sums), then tap K-1 on set a (c2 starts on set b)
    tap1(std::true_type{}, 0, wa);
    load_w(2, wa);
    tap1(std::false_type{}, 1, wb);
    load_w(3, wb);
#pragma unroll 1
    for (int g = 2; g + 1 < K; g += 2) {
      tap1(std::false_type{}, g, wa);
      load_w(g + 2, wa);
      tap1(std::false_type{}, g + 1, wb);
      load_w(g + 3, wb);
    }
    tap1(std::false_type{}, K - 1, wa);
    load_w(K + 1, wa);
  }
  TTSK_STAMP(3);
  // t = lrelu(c1 + b1) as fp16 rows of the t window; frames outside [0, len) are c2's zero padding
  // (a tile whose t window lies inside the utterance has no zero padding to write: no compare, no select — and the conversions pack in pairs)
  if (t0 - CP_TH >= 0 && t0 - CP_TH + CP_TROWS <= len) {
#pragma unroll
    for (int i = 0; i < NF1; ++i) {
      unsigned o[4];
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        f32x4 v = acc[cc][i] + bv1[cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
        o[cc * 2] = pack2<F16>(v[0], v[1]);
        o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
      }
      *(uint4*)(TW + (i * 16 + l15) * RS + co8 * 2) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NF1; ++i) {
      const int t = t0 - CP_TH + i * 16 + l15;
      const bool live = t >= 0 && t < len;
      unsigned o[4];
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        f32x4 v = acc[cc][i] + bv1[cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = live ? fmaxf(v[e], v[e] * a.slope) : 0.f;
        o[cc * 2] = pack2<F16>(v[0], v[1]);
        o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
      }
      *(uint4*)(TW + (i * 16 + l15) * RS + co8 * 2) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();

  TTSK_STAMP(4);
  // raw x (residual) and, in the accumulating modes, the current `out`: requested now, consumed after c2's taps
  uint4 rres[NF2], rout[NF2];
#pragma unroll
  for (int i = 0; i < NF2; ++i) {
    const int t = t0 + i * 16 + l15;
    rres[i] = rout[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < len) {
      rres[i] = *(const uint4*)(xb + (int64_t)t * C + co8);
      if (a.mode) rout[i] = *(const uint4*)(a.out + ((int64_t)bi * len + t) * C + co8);
    }
  }
  // ---- c2 (dilation 1) over the tile's 96 frames; sequence taps K .. 2K-1: tap K is on set b, K+1 on set a, ...
  f32x4 acc2[CT][NF2];
  {
    const unsigned char* inl = TW + (l15 + CP_TH) * RS + q * 16;
    auto tap2 = [&](auto firstc, int g, const bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {
      const unsigned char* inp = inl + (g - HK) * RS;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < NF2; ++i) {
          const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
          for (int cc = 0; cc < CT; ++cc) acc2[cc][i] = mfma16<F16>(w[ks][cc], Bf, (decltype(firstc)::value && ks == 0) ? zero4 : acc2[cc][i]);
        }
      }
    };
    tap2(std::true_type{}, 0, wb);
    load_w(K + 2, wb);
    tap2(std::false_type{}, 1, wa);
    if (3 < K) load_w(K + 3, wa);
#pragma unroll 1
    for (int g = 2; g + 1 < K; g += 2) {
      tap2(std::false_type{}, g, wb);
      if (g + 2 < K) load_w(K + g + 2, wb);
      tap2(std::false_type{}, g + 1, wa);
      if (g + 3 < K) load_w(K + g + 3, wa);
    }
    tap2(std::false_type{}, K - 1, wb);
  }

  TTSK_STAMP(5);
  // ---- epilogue: + b2 + raw x (+ the running sum), 16 bytes per frame tile and lane straight to memory
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
#pragma unroll
  for (int i = 0; i < NF2; ++i) {
    const int t = t0 + i * 16 + l15;
    const unsigned rw[4] = {rres[i].x, rres[i].y, rres[i].z, rres[i].w}, ow[4] = {rout[i].x, rout[i].y, rout[i].z, rout[i].w};
    unsigned o[4];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      f32x4 v = acc2[cc][i] + bv2[cc];
      {
        float r0, r1, r2, r3;
        unpack2<F16>(rw[cc * 2], r0, r1); unpack2<F16>(rw[cc * 2 + 1], r2, r3);
        v += f32x4{r0, r1, r2, r3};
      }
      if (a.mode) {
        {
          float o0, o1, o2, o3;
          unpack2<F16>(ow[cc * 2], o0, o1); unpack2<F16>(ow[cc * 2 + 1], o2, o3);
          v += f32x4{o0, o1, o2, o3};
        }
        if (a.mode == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= a.scale; v[e] = fmaxf(v[e], v[e] * a.final_slope); }
        }
      }
      o[cc * 2] = pack2<F16>(v[0], v[1]);
      o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
    }
    if (t < len) *(uint4*)(ob + (int64_t)t * C + co8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  TTSK_STAMP(6);
  TTSK_STAMP(7);
}

// ---- the same pair at C = 256 (the first MRF stage: 3,072 frames per utterance).  Eight waves, each 32 output channels x all
// frames; the 256 input channels of a tap are taken as two halves of 128 so that a wave's weights of one step stay 8 KiB = 32
// VGPRs (two register sets): the tap loop of the C = 128 kernel over 2K half-taps per conv.  LDS: x window 162 rows + t window
// 112 rows of 544 B = 149 KiB, one workgroup per CU; 32 tiles x 8 utterances = 256 workgroups = one per CU.  Replaces two grouped
// implicit-GEMM launches per dilation (48 KiB of operand fetch per 4.2 MFLOP there, 128 KiB of weights per tap and 96 frames
// = 25 MFLOP here).
constexpr int C256 = 256, C256_NW = 8, C256_NT = C256_NW * 64;
constexpr int C256_RS = C256 * 2 + 32;                               // 544 B per row
constexpr int C256_NC = C256 / 16, C256_KH = 4;                      // 16 cout tiles; 4 k-steps per 128-channel half
constexpr int C256_TAP = C256_NC * (C256 / 32) * 1024;               // 128 KiB per tap
constexpr int C256_SMEM = (CP_XROWS + CP_TROWS) * C256_RS;           // 149,056 B

template <bool F16>
__global__ __launch_bounds__(C256_NT, 1) void conv_pair256_kernel(const PairArgs a) {
  constexpr int C = C256, TT = CP_TT, RS = C256_RS, NC = C256_NC, KH = C256_KH, NT = C256_NT, CH8 = C / 8, CT = 2;
  constexpr int NF1 = CP_TROWS / 16, NF2 = TT / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C256_SMEM];
  unsigned char* XW = smem;
  unsigned char* TW = smem + CP_XROWS * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y, t0 = blockIdx.x * TT;
  const int len = a.len, K = a.K, d = a.dil;
  const int HK = (K - 1) / 2, K2 = 2 * K;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;

  // half-tap g of the sequence c1 (2K half-taps) | c2 (2K): tap (g mod 2K) / 2, channel half g & 1
  bf16x8 wa[KH][CT], wb[KH][CT];
  // (the A operand's rows permuted as in conv_pair_kernel: a lane's accumulator rows of its two tiles are 8 consecutive channels of one frame)
  int woff[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    const int co = 32 * wave + 8 * (l15 >> 2) + 4 * cc + (l15 & 3);
    woff[cc] = (co >> 4) * 1024 + ((co & 15) + 16 * q) * 16;
  }
  const int co8 = 32 * wave + 8 * q;
  auto load_w = [&](int g, bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
    const int gl = g < K2 ? g : g - K2;
    const unsigned char* src = (const unsigned char*)(g < K2 ? a.w1 : a.w2) + (int64_t)(gl >> 1) * C256_TAP + (gl & 1) * (KH * NC * 1024);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) w[ks][cc] = *(const bf16x8*)(src + ks * NC * 1024 + woff[cc]);
  };

  {  // ---- x window: lrelu(x) rows t0 - 33 .. t0 + 129, zeros outside the utterance.  The window is sized for dilation 5 (c1 reaches 25
     // rows beyond the 112 it computes); a smaller dilation reads HK * d rows either side only — the others are never fetched (they
     // stay zero, nobody reads them): 122 / 142 / 162 rows for d = 1 / 3 / 5 at k = 11.
    constexpr int NCH = (CP_XROWS * CH8 + NT - 1) / NT;     // 11
    const int xlo = CP_XH - CP_TH - HK * d, xhi = CP_XH - CP_TH + CP_TROWS + HK * d;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - CP_XH + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < CP_XROWS * CH8 && t >= 0 && t < len && row >= xlo && row < xhi) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
    // the weight fragments are requested BEHIND the window: loads return in order, and nothing starts before the window is in LDS
    load_w(0, wa);
    load_w(1, wb);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < CP_XROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = lrelu8_fast<F16>(xv[it], a.slope);
    }
  }
  f32x4 bv1[CT], bv2[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) {
    bv1[cc] = *(const f32x4*)(a.b1 + co8 + 4 * cc);
    bv2[cc] = *(const f32x4*)(a.b2 + co8 + 4 * cc);
  }
  __syncthreads();

  // ---- c1 over the 112 frames of the t window
  f32x4 acc[CT][NF1];      // (not zeroed: the first half-tap starts the sums, tapring.h FIRST)
  {
    const unsigned char* inl = XW + (l15 + CP_XH - CP_TH) * RS + q * 16;
    // (one 8-wave workgroup per CU: the two waves of a SIMD are in the same phase and hide nothing for each other — the activation
    // fragments go through tapring.h's ring, one k-step ahead of their MFMAs; the 2-workgroup pair kernels measured slower with it)
    auto inp1 = [&](int g) __attribute__((always_inline)) { return (unsigned)(inl - smem) + ((g >> 1) - HK) * d * RS + (g & 1) * (KH * 64); };
    bf16x8 ring[NF1];
    ring_prime<NF1, RS>(ring, smem, inp1(0));
    auto tap1 = [&](int g, const bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
      tap_ring<F16, KH, CT, NF1, RS>(acc, ring, w, smem, inp1(g), inp1(min(g + 1, K2 - 1)));
    };
    tap_ring<F16, KH, CT, NF1, RS, true>(acc, ring, wa, smem, inp1(0), inp1(1));
    load_w(2, wa);
    tap1(1, wb);
    load_w(3, wb);
#pragma unroll 1
    for (int g = 2; g < K2; g += 2) {      // the last two loads are c2's first two half-taps
      tap1(g, wa);
      load_w(g + 2, wa);
      tap1(g + 1, wb);
      load_w(g + 3, wb);
    }
  }
  if (t0 - CP_TH >= 0 && t0 - CP_TH + CP_TROWS <= len) {      // (the t window inside the utterance: no zero padding to write, no compare, no select)
#pragma unroll
    for (int i = 0; i < NF1; ++i) {
      unsigned o[4];
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        f32x4 v = acc[cc][i] + bv1[cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
        o[cc * 2] = pack2<F16>(v[0], v[1]);
        o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
      }
      *(uint4*)(TW + (i * 16 + l15) * RS + co8 * 2) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NF1; ++i) {
      const int t = t0 - CP_TH + i * 16 + l15;
      const bool live = t >= 0 && t < len;
      unsigned o[4];
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        f32x4 v = acc[cc][i] + bv1[cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = live ? fmaxf(v[e], v[e] * a.slope) : 0.f;
        o[cc * 2] = pack2<F16>(v[0], v[1]);
        o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
      }
      *(uint4*)(TW + (i * 16 + l15) * RS + co8 * 2) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();

  // raw x (residual): requested now, consumed after c2's taps (the accumulate operand of modes 1 / 2 is read in the epilogue:
  // its registers would spill here)
  uint4 rres[NF2];
#pragma unroll
  for (int i = 0; i < NF2; ++i) {
    const int t = t0 + i * 16 + l15;
    rres[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < len) rres[i] = *(const uint4*)(xb + (int64_t)t * C + co8);
  }
  // ---- c2 (dilation 1) over the tile's 96 frames
  f32x4 acc2[CT][NF2];
  {
    const unsigned char* inl = TW + (l15 + CP_TH) * RS + q * 16;
    auto inp2 = [&](int g) __attribute__((always_inline)) { return (unsigned)(inl - smem) + ((g >> 1) - HK) * RS + (g & 1) * (KH * 64); };
    bf16x8 ring[NF2];
    ring_prime<NF2, RS>(ring, smem, inp2(0));
    auto tap2 = [&](int g, const bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
      tap_ring<F16, KH, CT, NF2, RS>(acc2, ring, w, smem, inp2(g), inp2(min(g + 1, K2 - 1)));
    };
    tap_ring<F16, KH, CT, NF2, RS, true>(acc2, ring, wa, smem, inp2(0), inp2(1));
    load_w(K2 + 2, wa);
    tap2(1, wb);
    load_w(K2 + 3, wb);
#pragma unroll 1
    for (int g = 2; g < K2; g += 2) {
      tap2(g, wa);
      if (g + 2 < K2) load_w(K2 + g + 2, wa);
      tap2(g + 1, wb);
      if (g + 3 < K2) load_w(K2 + g + 3, wb);
    }
  }
  // ---- epilogue: + b2 + raw x (+ the running sum), 16 bytes per frame tile and lane straight to memory (no staging tile, no barrier)
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
#pragma unroll
  for (int i = 0; i < NF2; ++i) {
    const int t = t0 + i * 16 + l15;
    const unsigned rw[4] = {rres[i].x, rres[i].y, rres[i].z, rres[i].w};
    unsigned ow[4] = {0u, 0u, 0u, 0u};
    if (a.mode && t < len) {
      const uint4 ov = *(const uint4*)(ob + (int64_t)t * C + co8);
      ow[0] = ov.x; ow[1] = ov.y; ow[2] = ov.z; ow[3] = ov.w;
    }
    unsigned o[4];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      f32x4 v = acc2[cc][i] + bv2[cc];
      {
        float r0, r1, r2, r3;
        unpack2<F16>(rw[cc * 2], r0, r1); unpack2<F16>(rw[cc * 2 + 1], r2, r3);
        v += f32x4{r0, r1, r2, r3};
      }
      if (a.mode) {
        {
          float o0, o1, o2, o3;
          unpack2<F16>(ow[cc * 2], o0, o1); unpack2<F16>(ow[cc * 2 + 1], o2, o3);
          v += f32x4{o0, o1, o2, o3};
        }
        if (a.mode == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= a.scale; v[e] = fmaxf(v[e], v[e] * a.final_slope); }
        }
      }
      o[cc * 2] = pack2<F16>(v[0], v[1]);
      o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
    }
    if (t < len) *(uint4*)(ob + (int64_t)t * C + co8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ---- the same pair at C = 64 and C = 32 (the last two stages).  With 64 or 32 output channels there are not four 32-channel
// groups to give the waves, so the four waves split the 192 frames the workgroup computes for both convs as well: C = 64: two
// 32-channel groups x two 96-frame halves, C = 32: all channels x four 48-frame quarters; weights still L2 -> registers.
// c2 is computed on the same 192 rows as c1 (frames t0 - 8 .. t0 + 184) and only the middle 176 are stored: the t window
// carries 8 guard rows each side that only those discarded edge rows read.  Against the six-conv fused kernel
// (resblock.hip): that one reads x once per block but multiplies 1.23-1.45x redundant halo rows and synchronises its eight
// waves per weight stage; three pair launches move 3x its HBM bytes (still one read + one write of x per pair) and win on the
// MFMA side.
template <int C, int NW_> struct FsGeom {
  static constexpr int NW = NW_, NT = NW * 64, GU = 8;
  // wave = (cout group of 32 channels, frame group): C = 64, 4 waves: 2 x 2 — 32 output channels x 96 frames per wave (6 MFMAs per weight
  // fragment loaded, the weights of a tap fetched twice per workgroup); all 64 channels x 48 frames per wave measured 29 % MFMA busy (3
  // MFMAs per fragment, four fetches).  C = 32: one cout group, four frame groups of 48.  (Round 4 measured 8-wave variants — C = 128:
  // 4 cout groups x 2 frame halves, C = 64: 2 x 4, one workgroup per CU, twice the frames behind each weight fetch: 1.07 vs 0.87 ms and
  // 0.66 vs 0.60 ms per stage.  Two independent 4-wave workgroups per CU in different phases beat one 8-wave workgroup in lockstep.)
  static constexpr int CG = C >= 64 ? C / 32 : 1, FG = NW / CG, FW = C >= 64 ? 96 : 48, NF = FW / 16;
  static constexpr int CROWS = FG * FW;                 // rows computed by both convs
  static constexpr int TT = CROWS - 2 * GU;             // frames stored
  static constexpr int XROWS = CROWS + 50;              // c1 reaches 25 rows either side
  static constexpr int TROWS = CROWS + 2 * GU;
  // Row stride C * 2 + 32 bytes = 2 mod 4 sixteen-byte units, like the C = 128 / 256 kernels: a ds_read_b128 is served in groups of 16
  // lanes that mix 8 rows at k-chunk q with the 8 OTHER rows at q + 1; with an odd stride in units (C * 2 + 16: what this kernel had)
  // rows l and l' with 9 l = 9 l' + 1 (mod 16) collide for every such pair — every fragment read two-way conflicted, 48 % of the
  // kernel's LDS cycles (profiles/r03_mfma_util.json).  A stride of 2 mod 4 units sends q-even chunks to even units and q-odd ones to
  // odd units, 8 rows each over the 8 units of a parity: conflict-free at every tap shift.
  static constexpr int RS = C * 2 + 32;
  static constexpr int NC = C / 16, KS = C / 32, CH8 = C / 8;
  static constexpr int TAP = NC * KS * 1024;
  static constexpr int SMEM = (XROWS + TROWS) * RS;     // 72,000 B (C = 64, 4 waves: two workgroups per CU) / 43,200 B (C = 32)
  static constexpr int OCC = SMEM <= 80 * 1024 ? 2 : 1; // workgroups per CU the LDS allows (at most two are asked for)
};

template <int C, int NW, bool F16>
__global__ __launch_bounds__(NW * 64, (FsGeom<C, NW>::OCC * NW) / 4) void conv_pair_fs_kernel(const PairArgs a) {
  using Gm = FsGeom<C, NW>;
  constexpr int NT = Gm::NT, FW = Gm::FW, NF = Gm::NF, GU = Gm::GU, TT = Gm::TT, XROWS = Gm::XROWS, RS = Gm::RS, NC = Gm::NC, KS = Gm::KS,
                CH8 = Gm::CH8, TAP = Gm::TAP, XH = GU + 25, CG = Gm::CG, CT = NC / CG;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Gm::SMEM];
  unsigned char* XW = smem;
  unsigned char* TW = smem + XROWS * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cgi = wave % CG, fg = wave / CG;             // this wave's output-channel group and frame group
  const int bi = blockIdx.y, t0 = blockIdx.x * TT;
  const int len = a.len, K = a.K, d = a.dil;
  const int HK = (K - 1) / 2;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;

  bf16x8 wa[KS][CT], wb[KS][CT];
  auto load_w = [&](int g, bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {      // tap g of the 2K-tap sequence c1 | c2
    const unsigned char* src = (const unsigned char*)(g < K ? a.w1 : a.w2) + (int64_t)(g < K ? g : g - K) * TAP + (cgi * CT) * 1024 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int c = 0; c < CT; ++c) w[ks][c] = *(const bf16x8*)(src + (ks * NC + c) * 1024);
  };

  {  // ---- x window: lrelu(x) rows t0 - 33 .. t0 + 209, zeros outside the utterance
    constexpr int NCH = (XROWS * CH8 + NT - 1) / NT;
    const int xlo = 25 - HK * d, xhi = 25 + Gm::CROWS + HK * d;      // the rows c1 reads at this dilation (the window is sized for d = 5)
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int t = t0 - XH + row;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < XROWS * CH8 && t >= 0 && t < len && row >= xlo && row < xhi) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
    // the weight fragments are requested BEHIND the window: loads return in order, and nothing starts before the window is in LDS
    load_w(0, wa);
    load_w(1, wb);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < XROWS * CH8) *(uint4*)(XW + row * RS + ch * 16) = lrelu8_fast<F16>(xv[it], a.slope);
    }
  }
  f32x4 bv1[CT], bv2[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    bv1[c] = *(const f32x4*)(a.b1 + (cgi * CT + c) * 16 + q * 4);
    bv2[c] = *(const f32x4*)(a.b2 + (cgi * CT + c) * 16 + q * 4);
  }
  __syncthreads();

  f32x4 acc[CT][NF];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // one tap of either conv: `inl` = this lane's fragment address at shift 0, rows advance by `step` rows per tap offset
  auto tap = [&](const unsigned char* inp, const bf16x8 (&w)[KS][CT]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + ks * 64);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][i] = mfma16<F16>(w[ks][c], Bf, acc[c][i]);
      }
    }
  };

  // ---- c1: computed row r = wave*48 + i*16 + l15 <-> frame t0 - 8 + r <-> x-window row r + 25
  zero_acc();
  {
    const unsigned char* inl = XW + (fg * FW + l15 + 25) * RS + q * 16;
#pragma unroll 1
    for (int g = 0; g + 1 < K; g += 2) {
      tap(inl + (g - HK) * d * RS, wa);
      load_w(g + 2, wa);
      tap(inl + (g + 1 - HK) * d * RS, wb);
      load_w(g + 3, wb);
    }
    tap(inl + (K - 1 - HK) * d * RS, wa);
    load_w(K + 1, wa);
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int r = fg * FW + i * 16 + l15;
    const int t = t0 - GU + r;
    const bool live = t >= 0 && t < len;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      f32x4 v = acc[c][i] + bv1[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = live ? fmaxf(v[e], v[e] * a.slope) : 0.f;
      *(uint2*)(TW + (r + GU) * RS + ((cgi * CT + c) * 16 + q * 4) * 2) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
    }
  }
  __syncthreads();

  // raw x (residual) and, in the accumulating modes, the current `out`: requested now, consumed after c2's taps
  uint2 rres[CT][NF], rout[CT][NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int r = fg * FW + i * 16 + l15;
    const int t = t0 - GU + r;
    const bool mine = r >= GU && r < GU + TT && t < len;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      rres[c][i] = rout[c][i] = make_uint2(0u, 0u);
      if (mine) {
        rres[c][i] = *(const uint2*)(xb + (int64_t)t * C + (cgi * CT + c) * 16 + q * 4);
        if (a.mode) rout[c][i] = *(const uint2*)(a.out + ((int64_t)bi * len + t) * C + (cgi * CT + c) * 16 + q * 4);
      }
    }
  }
  // ---- c2 on the same rows (the 8 at either end read guard rows and are dropped)
  zero_acc();
  {
    const unsigned char* inl = TW + (fg * FW + l15 + GU) * RS + q * 16;
#pragma unroll 1
    for (int g = 0; g + 1 < K; g += 2) {
      tap(inl + (g - HK) * RS, wb);
      if (g + 2 < K) load_w(K + g + 2, wb);
      tap(inl + (g + 1 - HK) * RS, wa);
      if (g + 3 < K) load_w(K + g + 3, wa);
    }
    tap(inl + (K - 1 - HK) * RS, wb);
  }

  // ---- epilogue: + b2 + raw x, MRF mode, staged through the x window for full-row stores
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int r = fg * FW + i * 16 + l15;
    const int t = t0 - GU + r;
    const bool mine = r >= GU && r < GU + TT && t < len;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int co = (cgi * CT + c) * 16 + q * 4;
      f32x4 v = acc[c][i] + bv2[c];
      {
        float r0, r1, r2, r3;
        unpack2<F16>(rres[c][i].x, r0, r1); unpack2<F16>(rres[c][i].y, r2, r3);
        v += f32x4{r0, r1, r2, r3};
      }
      if (a.mode) {
        {
          float o0, o1, o2, o3;
          unpack2<F16>(rout[c][i].x, o0, o1); unpack2<F16>(rout[c][i].y, o2, o3);
          v += f32x4{o0, o1, o2, o3};
        }
        if (a.mode == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= a.scale; v[e] = fmaxf(v[e], v[e] * a.final_slope); }
        }
      }
      if (r >= GU && r < GU + TT) *(uint2*)(XW + (r - GU) * RS + co * 2) = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
    }
  }
  __syncthreads();
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
  constexpr int NCO = (TT * CH8 + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < NCO; ++it) {
    const int idx = it * NT + tid;
    const int rr = idx / CH8, ch = idx - rr * CH8;
    const int t = t0 + rr;
    if (idx < TT * CH8 && t < len) *(uint4*)(ob + (int64_t)t * C + ch * 8) = *(const uint4*)(XW + rr * RS + ch * 16);
  }
}


}  // namespace

extern "C" int ttsk_hifi_conv_window_supported(int C, int K, int dil) {
  return C == CW_C && K >= 1 && K <= 11 && (K & 1) == 1 && dil >= 1 && dil * ((K - 1) / 2) <= C2_H;
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
  dim3 grid((len + 191) / 192, B);
  if (f16) hipLaunchKernelGGL((conv_window2_kernel<true, 192, 2>), grid, dim3(C2_NT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv_window2_kernel<false, 192, 2>), grid, dim3(C2_NT), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

#ifdef TTSK_STAMPS
static unsigned long long* g_pair_stamps = nullptr;
// diagnostic build only (`make stamps` -> libttsk_hip_stamps.so, tools/debug/pair_stamps.py; not declared in ttsk.h, not in the
// product library): device buffer of 8 x uint64 per workgroup of the C = 128 kernel for the launches that follow; null switches
// the stamps off again
extern "C" int ttsk_hifi_conv_pair_set_stamps(void* dev_buffer) {
  g_pair_stamps = (unsigned long long*)dev_buffer;
  return TTSK_OK;
}
#endif

extern "C" int ttsk_hifi_conv_pair_supported(int C, int K, int dil) {
  return (C == CW_C || C == C256 || C == 64 || C == 32) && K >= 3 && K <= 11 && (K & 1) == 1 && dil >= 1 && dil * ((K - 1) / 2) <= CP_XH - CP_TH && (K - 1) / 2 <= CP_TH;
}

extern "C" int ttsk_hifi_conv_pair(const void* x16, const void* w1_pack, const float* bias1, const void* w2_pack, const float* bias2,
                                   void* out16, int f16, int B, int len, int C, int K, int dil, float slope, int mode, float scale,
                                   float final_slope, void* stream) {
  TTSK_REQUIRE(x16 && w1_pack && bias1 && w2_pack && bias2 && out16, "ttsk_hifi_conv_pair: null pointer");
  TTSK_REQUIRE(B > 0 && len > 0 && B <= 65535 && x16 != out16, "ttsk_hifi_conv_pair: bad sizes / in-place output");
  TTSK_REQUIRE(ttsk_hifi_conv_pair_supported(C, K, dil), "ttsk_hifi_conv_pair: no instance for C=%d K=%d dil=%d", C, K, dil);
  TTSK_REQUIRE(mode >= 0 && mode <= 2 && (final_slope > 0.f || mode != 2), "ttsk_hifi_conv_pair: bad mode / final_slope");
  TTSK_REQUIRE(slope > 0.f && slope < 1.f, "ttsk_hifi_conv_pair: LeakyReLU slope %g outside (0, 1)", slope);
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w1_pack) | ((uintptr_t)w2_pack) | ((uintptr_t)bias1) | ((uintptr_t)bias2) | ((uintptr_t)out16)) & 15) == 0,
               "ttsk_hifi_conv_pair: 16-byte alignment");
  PairArgs a{(const bf16_t*)x16, (const bf16_t*)w1_pack, (const bf16_t*)w2_pack, bias1, bias2, (bf16_t*)out16, len, K, dil, slope,
             mode, scale, final_slope};
#ifdef TTSK_STAMPS
  a.stamps = g_pair_stamps;
#endif
  if (C == C256) {
    dim3 grid((len + CP_TT - 1) / CP_TT, B);
    if (f16) hipLaunchKernelGGL(conv_pair256_kernel<true>, grid, dim3(C256_NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_pair256_kernel<false>, grid, dim3(C256_NT), 0, (hipStream_t)stream, a);
  } else if (C == CW_C) {
    dim3 grid((len + CP_TT - 1) / CP_TT, B);
    if (f16) hipLaunchKernelGGL(conv_pair_kernel<true>, grid, dim3(C2_NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_pair_kernel<false>, grid, dim3(C2_NT), 0, (hipStream_t)stream, a);
  } else {
    dim3 grid((len + FsGeom<64, 4>::TT - 1) / FsGeom<64, 4>::TT, B);
    if (C == 64) {
      if (f16) hipLaunchKernelGGL((conv_pair_fs_kernel<64, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((conv_pair_fs_kernel<64, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
      if (f16) hipLaunchKernelGGL((conv_pair_fs_kernel<32, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((conv_pair_fs_kernel<32, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
  }
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
