// resblock.hip — HiFi-GAN ResBlock1, all six convolutions of the block fused into one kernel per time tile.
// reference: hifi/models.py:12-95 (ResBlock1.forward :88-95), called from Generator.forward :190-196.
//
//   for m in 0..2:   t = conv_{k, dil d_m}(lrelu(x)) ;  x = conv_{k, dil 1}(lrelu(t)) + x
//
// One 256-thread workgroup owns TT output frames of one utterance plus the halo the six convolutions consume
// (H = (k-1)/2 * (d0+d1+d2+3) frames per side).  The tile lives in LDS for the whole block: XL = lrelu(x) and
// TL = lrelu(t) as 16-bit [frame][channel] rows (16-byte chunks XOR-swizzled by the frame: conflict-free ds_read_b128
// fragments at every tap shift); the residual
// x itself stays in fp32 REGISTERS in MFMA accumulator layout, so HBM sees one read of x and one write of the result
// per ResBlock instead of 5 tensor passes per conv pair.
//
// Each conv is D[cout][frame] = sum_tap sum_cin W[cout][tap][cin] * IN[frame + (tap-(k-1)/2)*d][cin] on
// v_mfma_f32_16x16x32: the activation fragment (B operand) is one ds_read_b128 of 16 frames x 8 channels shifted by the
// tap and is reused for every 16-channel output tile.  Frames on the MFMA column axis make every lane own 4 consecutive
// output channels of one frame: epilogues write 8-byte channel runs.  A wave owns frame tiles (wave, wave+4, ...) for
// ALL convs, so the residual of an element is always in the lane that produces its update.
//
// Weights: every wave needs every weight of the conv, and 4 waves x C*C*2 B per tap straight from L1 saturate the CU's
// 64 B/clk vector-memory path (first version of this kernel: 13 % MFMA utilisation).  So the weights are pre-packed in
// MFMA-fragment order ([tap][k-step][cout-tile][lane][8], ttsk_pack_resblock_weight) and streamed through a double-
// buffered LDS stage of G taps: the workgroup fetches each stage once (coalesced 16-B loads into registers, requested
// FOUR stages ahead — every CU reads the same few KB at the same time, and the resulting L2 latency of ~2k cycles is
// longer than one stage of MFMAs — written to LDS one stage ahead), one barrier per stage, and each wave reads its A
// fragments with conflict-free lane-linear ds_read_b128.  8 waves per workgroup (2 per SIMD) overlap one wave's LDS
// latency and epilogue with the other's MFMAs.
#include <type_traits>
#include "common.h"

namespace {

struct RbArgs {
  const bf16_t* x;
  bf16_t* out;
  const bf16_t* w[6];   // convs1[0], convs2[0], convs1[1], convs2[1], convs1[2], convs2[2]: fragment-major packs
  const float* b[6];
  int dil[3];
  int len;              // frames per utterance
  int mode;             // 0: out = y   1: out += y   2: out = (out + y) * scale
  float scale;
  float slope;          // LeakyReLU slope inside the block (0.1)
  float final_slope;    // LeakyReLU applied to the stored value (1.0 = none): the consumer's activation, fused
};

template <int C, int K, int TT, int NW>
struct RbGeom {
  static constexpr int NT = NW * 64;                  // threads per workgroup
  static constexpr int HK = (K - 1) / 2;
  static constexpr int H = 6 * (K - 1);               // halo per side for dilations (1,3,5): HK * (1+3+5+3)
  static constexpr int ROWS = TT + 2 * H;
  static constexpr int NTILE = (ROWS + 15) / 16;
  static constexpr int NSLOT = (NTILE + NW - 1) / NW;
  static constexpr int G = 32;                         // guard rows either side (>= max tap reach 5*HK = 25)
  static constexpr int LROWS = NSLOT * NW * 16 + 2 * G;   // every multiplied tile (also the unused ones past NTILE) reads in-bounds
  static constexpr int RS = C * 2;                     // row stride in bytes; 16-byte chunks are XOR-swizzled by the row
  static constexpr int NC = C / 16;
  static constexpr int KS = C / 32;
  static constexpr int TAP_BYTES = NC * KS * 1024;     // one tap of weights in fragment order
  static constexpr int GT = C == 32 ? K : (TT > 136 ? 2 : 4);   // taps per weight stage (C = 64: 2 when the 3-slot tile needs the LDS)
  static constexpr int NS = (K + GT - 1) / GT;         // stages per conv
  static constexpr int WSTAGE = GT * TAP_BYTES;
  static constexpr int NLD = (WSTAGE + NT * 16 - 1) / (NT * 16);   // 16-byte loads per thread per stage
  static constexpr int ACT = 2 * LROWS * RS;
  static constexpr int SMEM = ACT + 2 * WSTAGE;
};

template <int C, int K, int TT, int NW, bool F16>
__global__ __launch_bounds__(NW * 64, 1) void resblock1_kernel(const RbArgs a) {
  using Gm = RbGeom<C, K, TT, NW>;
  constexpr int NT = Gm::NT;
  constexpr int HK = Gm::HK, H = Gm::H, NTILE = Gm::NTILE, NSLOT = Gm::NSLOT, G = Gm::G, LROWS = Gm::LROWS, RS = Gm::RS,
                NC = Gm::NC, KS = Gm::KS, CH8 = C / 8, GT = Gm::GT, NS = Gm::NS, WSTAGE = Gm::WSTAGE, NLD = Gm::NLD,
                TAP_BYTES = Gm::TAP_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Gm::SMEM];
  unsigned char* XL = smem;
  unsigned char* TL = smem + LROWS * RS;
  unsigned char* WB = smem + Gm::ACT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int bi = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int len = a.len;
  const bf16_t* __restrict__ xb = a.x + (int64_t)bi * len * C;
  const float slope = a.slope;
  // chunk swizzle of the activation rows: conflict-free ds_read_b128 fragments for any tap shift (C = 64: 8 chunks per
  // row, ^ row & 7; C = 32: 4 chunks per row, ^ (row >> 1) & 3); tile bases and the guard are multiples of 8 rows
  auto swz = [](int row) __attribute__((always_inline)) { return C == 64 ? (row & 7) : ((row >> 1) & 3); };

  // ---- weight stage gs (global index over the 6 convs): conv gs / NS, taps (gs % NS)*GT .. +GT of its pack
  constexpr int NSET = 4;   // register sets of weight stages in flight: stage g travels in set g % NSET
  uint4 wr0[NLD], wr1[NLD], wr2[NLD], wr3[NLD];
  auto load_stage = [&](int gs, uint4 (&wr)[NLD]) __attribute__((always_inline)) {
    const int conv = gs / NS, si = gs - conv * NS;
    // packs are padded to NS * GT taps (zeros), so a short last stage loads in-bounds and no select touches the
    // loaded value: the loads stay in flight until store_stage
    const unsigned char* src = (const unsigned char*)a.w[conv] + (int64_t)si * WSTAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int off = (i * NT + tid) * 16;
      if ((i + 1) * NT * 16 <= WSTAGE || off < WSTAGE) wr[i] = *(const uint4*)(src + off);
    }
  };
  auto store_stage = [&](int gs, uint4 (&wr)[NLD]) __attribute__((always_inline)) {
    unsigned char* dst = WB + (gs & 1) * WSTAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int off = (i * NT + tid) * 16;
      if ((i + 1) * NT * 16 <= WSTAGE || off < WSTAGE) *(uint4*)(dst + off) = wr[i];
    }
  };
#pragma unroll
  for (int i = 0; i < NLD; ++i) wr0[i] = wr1[i] = wr2[i] = wr3[i] = make_uint4(0, 0, 0, 0);
  load_stage(0, wr0);
  if (6 * NS > 1) load_stage(1, wr1);
  if (6 * NS > 2) load_stage(2, wr2);
  if (6 * NS > 3) load_stage(3, wr3);

  // ---- stage the tile: TL <- raw x (only to initialise the residual registers), XL <- lrelu(x); zeros outside
  //      the utterance (conv zero padding) and in the guard rows.  All of a thread's loads are issued before the first
  //      one is used: one HBM latency per tile instead of one per 16-byte chunk.
  {
    constexpr int NCH = (LROWS * CH8 + NT - 1) / NT;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      const int r = row - G;
      const int t = t0 - H + r;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < LROWS * CH8 && r >= 0 && r < NTILE * 16 && t >= 0 && t < len) xv[it] = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    }
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
  store_stage(0, wr0);
  if (6 * NS > NSET) load_stage(NSET, wr0);
  __syncthreads();

  // ---- residual registers: xr[c][s] = x[frame = tile(s)*16 + l15][channels c*16 + q*4 .. +3]
  f32x4 xr[NC][NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int i = s * NW + wave;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint2 v = make_uint2(0, 0);
      if (i < NTILE) v = *(const uint2*)(TL + (i * 16 + l15 + G) * RS + (((c * 2 + (q >> 1)) ^ swz(l15)) << 4) + (q & 1) * 8);
      float x0, x1, x2, x3;
      unpack2<F16>(v.x, x0, x1); unpack2<F16>(v.y, x2, x3);
      xr[c][s] = f32x4{x0, x1, x2, x3};
    }
  }
  __syncthreads();

  int S = 0;   // halo consumed so far: conv outputs are needed (and valid) on rows [S, ROWS - S)
  int gs = 0;  // weight stage counter over the whole block (buffer = gs & 1)
#pragma unroll 1
  for (int conv = 0; conv < 6; ++conv) {
    const int half = conv & 1;
    const int d = half == 0 ? a.dil[conv >> 1] : 1;
    S += HK * d;
    const int tlo = S >> 4, thi = (Gm::ROWS - S + 15) >> 4;
    const unsigned char* in = half == 0 ? XL : TL;
    const unsigned char* inl = in + (l15 + G) * RS;

    f32x4 acc[NC][NSLOT];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) acc[c][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias of this conv: requested now, needed only by the epilogue (keeps a global-load latency off its critical path)
    f32x4 bv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) bv[c] = *(const f32x4*)(a.b[conv] + c * 16 + q * 4);

#pragma unroll 1
    for (int si = 0; si < NS; ++si, ++gs) {
      const unsigned char* wb = WB + (gs & 1) * WSTAGE + lane * 16;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        const int tap = si * GT + g;
        if (tap < K) {
          bf16x8 Af[NC * KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int c = 0; c < NC; ++c) Af[c * KS + ks] = *(const bf16x8*)(wb + ((g * KS + ks) * NC + c) * 1024);
          const int shift = (tap - HK) * d;
          const unsigned char* inp = inl + shift * RS;
          const int sw = swz(l15 + shift + 64);
          // every slot is multiplied, also tiles outside [tlo, thi): their rows read in-bounds (guard rows) garbage
          // that no valid output depends on, and straight-line code lets the LDS reads run ahead of the MFMAs
#pragma unroll
          for (int s = 0; s < NSLOT; ++s) {
            const int i = s * NW + wave;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const bf16x8 Bf = *(const bf16x8*)(inp + i * 16 * RS + (((ks * 4 + q) ^ sw) << 4));
#pragma unroll
              for (int c = 0; c < NC; ++c) acc[c][s] = mfma16<F16>(Af[c * KS + ks], Bf, acc[c][s]);
            }
          }
        }
      }
      // stage gs+1 (register set (gs+1) % NSET, requested NSET stages ago) goes to the LDS buffer stage gs-1 was read
      // from; the freed register set requests stage gs+1+NSET
      switch ((gs + 1) & (NSET - 1)) {
        case 0: if (gs + 1 < 6 * NS) store_stage(gs + 1, wr0); if (gs + 1 + NSET < 6 * NS) load_stage(gs + 1 + NSET, wr0); break;
        case 1: if (gs + 1 < 6 * NS) store_stage(gs + 1, wr1); if (gs + 1 + NSET < 6 * NS) load_stage(gs + 1 + NSET, wr1); break;
        case 2: if (gs + 1 < 6 * NS) store_stage(gs + 1, wr2); if (gs + 1 + NSET < 6 * NS) load_stage(gs + 1 + NSET, wr2); break;
        default: if (gs + 1 < 6 * NS) store_stage(gs + 1, wr3); if (gs + 1 + NSET < 6 * NS) load_stage(gs + 1 + NSET, wr3); break;
      }
      if (si + 1 < NS) __syncthreads();
    }

    // ---- conv epilogue
    const bool last = conv == 5;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const int i = s * NW + wave;
        if (i >= tlo && i < thi) {
          const int t = t0 - H + i * 16 + l15;
          const float okf = (t >= 0 && t < len) ? 1.f : 0.f;     // positions outside the utterance are conv zero padding
          f32x4 v = acc[c][s] + bv[c];
          unsigned char* dst;
          if (half == 0) {
            dst = TL;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope) * okf;
          } else {
            dst = XL;
            v = (v + xr[c][s]) * okf;
            xr[c][s] = v;
            if (!last) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
            }
          }
          *(uint2*)(dst + (i * 16 + l15 + G) * RS + (((c * 2 + (q >> 1)) ^ swz(l15)) << 4) + (q & 1) * 8) =
              make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
        }
      }
    }
    __syncthreads();
  }

  // ---- XL rows [H, H+TT) now hold the block output: coalesced 16-byte copy-out (+ MRF sum / average / activation)
  bf16_t* __restrict__ ob = a.out + (int64_t)bi * len * C;
  const float fsl = a.final_slope;
  constexpr int NCO = (TT * CH8 + NT - 1) / NT;
  uint4 prev[NCO];
#pragma unroll
  for (int it = 0; it < NCO; ++it) {        // reads of the running MRF sum first, all in flight together
    const int idx = it * NT + tid;
    const int rr = idx / CH8, ch = idx - rr * CH8;
    const int t = t0 + rr;
    prev[it] = make_uint4(0, 0, 0, 0);
    if (a.mode != 0 && idx < TT * CH8 && t < len) prev[it] = *(const uint4*)(ob + (int64_t)t * C + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < NCO; ++it) {
    const int idx = it * NT + tid;
    const int rr = idx / CH8, ch = idx - rr * CH8;
    const int t = t0 + rr;
    if (idx >= TT * CH8 || t >= len) continue;
    uint4 v = *(const uint4*)(XL + (rr + H + G) * RS + ((ch ^ swz(rr + H + G)) << 4));
    if (a.mode != 0 || fsl != 1.f) {
      const uint4 o = prev[it];
      const unsigned vw[4] = {v.x, v.y, v.z, v.w}, ow[4] = {o.x, o.y, o.z, o.w};
      unsigned rw[4];
      const float sc = a.mode == 2 ? a.scale : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float vl, vh, ol, oh;
        unpack2<F16>(vw[e], vl, vh); unpack2<F16>(ow[e], ol, oh);
        vl = (vl + ol) * sc; vh = (vh + oh) * sc;
        vl = vl > 0.f ? vl : vl * fsl; vh = vh > 0.f ? vh : vh * fsl;
        rw[e] = pack2<F16>(vl, vh);
      }
      v = make_uint4(rw[0], rw[1], rw[2], rw[3]);
    }
    *(uint4*)(ob + (int64_t)t * C + ch * 8) = v;
  }
}

// (Cout, Cin, K) fp32 -> fragment-major 16-bit pack [Kpad][C/32][C/16][64 lanes][8], Kpad = K rounded up to the weight
// stage (zeros): lane l of fragment (tap, ks, c) holds W[c*16 + (l & 15)][ks*32 + (l >> 4)*8 + j][tap], j = 0..7
template <bool F16>
__global__ __launch_bounds__(256) void pack_rb_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int C, int K,
                                                             int Kpad) {
  const int64_t n = (int64_t)C * C * Kpad;
  const int NC = C / 16, KS = C / 32;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i & 7);
    const int l = (int)((i >> 3) & 63);
    int f = (int)(i >> 9);
    const int c = f % NC; f /= NC;
    const int ks = f % KS;
    const int tap = f / KS;
    const int co = c * 16 + (l & 15), ci = ks * 32 + (l >> 4) * 8 + j;
    dst[i] = tap < K ? pack1<F16>(src[((int64_t)co * C + ci) * K + tap]) : (bf16_t)0;
  }
}

int rb_kpad(int C, int K) {
  const int gt = C == 32 ? K : (C == 64 ? 4 : 1);      // taps per weight stage of the consuming kernel
  return (K + gt - 1) / gt * gt;
}

template <int C, int K, int TT, int NW>
int launch_rb(const RbArgs& a, int B, int f16, hipStream_t s) {
  dim3 grid((a.len + TT - 1) / TT, B);
  if (f16) hipLaunchKernelGGL((resblock1_kernel<C, K, TT, NW, true>), grid, dim3(NW * 64), 0, s, a);
  else hipLaunchKernelGGL((resblock1_kernel<C, K, TT, NW, false>), grid, dim3(NW * 64), 0, s, a);
  return 0;
}

}  // namespace

extern "C" int64_t ttsk_resblock_pack_elems(int C, int K) { return (int64_t)C * C * rb_kpad(C, K); }

extern "C" int ttsk_pack_resblock_weight(const float* src, void* dst16, int f16, int C, int K, void* stream) {
  TTSK_REQUIRE(src && dst16 && (C == 32 || C == 64 || C == 128 || C == 256) && K >= 1, "ttsk_pack_resblock_weight: C must be 32, 64, 128 or 256");
  const int Kpad = rb_kpad(C, K);
  const int64_t n = (int64_t)C * C * Kpad;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (f16) hipLaunchKernelGGL(pack_rb_weight_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst16, C, K, Kpad);
  else hipLaunchKernelGGL(pack_rb_weight_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst16, C, K, Kpad);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_hifi_resblock1(const void* x16, void* out16, int f16, const void* const* weights /* 6 fragment-major packs */,
                                   const float* const* biases /* 6 x [C] */, const int32_t* dilations /* 3 */, int B, int len,
                                   int C, int K, int mode, float scale, float slope, float final_slope, void* stream) {
  TTSK_REQUIRE(x16 && out16 && weights && biases && dilations, "ttsk_hifi_resblock1: null pointer");
  TTSK_REQUIRE(B > 0 && len > 0 && B <= 65535, "ttsk_hifi_resblock1: bad sizes B=%d len=%d", B, len);
  TTSK_REQUIRE(mode >= 0 && mode <= 2, "ttsk_hifi_resblock1: mode");
  TTSK_REQUIRE(slope > 0.f && slope < 1.f, "ttsk_hifi_resblock1: LeakyReLU slope %g outside (0, 1)", slope);
  TTSK_REQUIRE(dilations[0] >= 1 && dilations[1] >= 1 && dilations[2] >= 1 && dilations[0] + dilations[1] + dilations[2] <= 9 &&
                   dilations[0] <= 5 && dilations[1] <= 5 && dilations[2] <= 5,
               "ttsk_hifi_resblock1: dilations (%d,%d,%d) exceed the tile halo built for (1,3,5)", dilations[0], dilations[1], dilations[2]);
  RbArgs a;
  a.x = (const bf16_t*)x16;
  a.out = (bf16_t*)out16;
  for (int i = 0; i < 6; ++i) {
    TTSK_REQUIRE(weights[i] && biases[i], "ttsk_hifi_resblock1: null weight/bias %d", i);
    TTSK_REQUIRE((((uintptr_t)weights[i]) & 15) == 0 && (((uintptr_t)biases[i]) & 15) == 0, "ttsk_hifi_resblock1: 16-byte alignment");
    a.w[i] = (const bf16_t*)weights[i];
    a.b[i] = biases[i];
  }
  TTSK_REQUIRE((((uintptr_t)x16) & 15) == 0 && (((uintptr_t)out16) & 15) == 0, "ttsk_hifi_resblock1: 16-byte alignment");
  for (int i = 0; i < 3; ++i) a.dil[i] = dilations[i];
  a.len = len; a.mode = mode; a.scale = scale; a.slope = slope; a.final_slope = final_slope;
  hipStream_t s = (hipStream_t)stream;
  const int key = C * 100 + K;
  switch (key) {
    // C = 64: TT = 3 frame tiles per wave * 8 waves * 16 - 2 * halo, so every multiplied tile is a needed one (384 rows,
    // 147 KB of LDS; K = 11: 494 -> 401 us against TT = 128, whose 248-row tile multiplied 256).  C = 32 measured slower
    // with exact-fit tiles (360 / 568 / 520: 111 / 151 / 188 us against 82 / 137 / 176) and keeps the power-of-two ones.
    case 3203: launch_rb<32, 3, 256, 8>(a, B, f16, s); break;
    case 3207: launch_rb<32, 7, 512, 8>(a, B, f16, s); break;
    case 3211: launch_rb<32, 11, 512, 8>(a, B, f16, s); break;
    case 6403: launch_rb<64, 3, 360, 8>(a, B, f16, s); break;
    case 6407: launch_rb<64, 7, 312, 8>(a, B, f16, s); break;
    case 6411: launch_rb<64, 11, 264, 8>(a, B, f16, s); break;
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
