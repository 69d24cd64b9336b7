// pairws.hip — HiFi-GAN's (c1 dilated -> LeakyReLU -> c2 -> + x) pair of ResBlock1 at C = 64 with the WEIGHTS STATIONARY in registers.
// reference: hifi/models.py:88-95 (ResBlock1.forward), called from Generator.forward :190-196.
//
// What convwin.hip's pair kernels pay per 176-frame workgroup at C = 64 (round 5, profiles/r05_mfma_util.json: MFMA busy 0.31): every
// wave streams its own copy of both convs' weights L2 -> registers (229 KB per workgroup at k = 7: as long as its MFMAs), the x
// window, the two convs and the stores run one after the other, and two such workgroups per CU do not cover each other's phases: a
// launch costs (HBM time) + (weight-stream time) + (MFMA time).  With 64 channels a conv's weights are small — 32 output channels x
// K taps x 64 input channels are 16 K registers per lane (176 at k = 11) — so here they never move again after the first
// microsecond:
//   * one persistent 8-wave workgroup per CU walks a contiguous run of 192-frame tiles;
//   * waves 0-3 hold c1's weights, waves 4-7 c2's (wave = 32 output channels x one half of the tile's frames); the c1 waves compute
//     t = lrelu(c1(lrelu(x))) of tile s into one LDS window while the c2 waves turn tile s-1's window into its output: the two waves
//     of a SIMD are in different phases BY CONSTRUCTION (one's epilogue and memory traffic under the other's MFMAs), and ONE
//     workgroup barrier per tile is all the synchronisation there is;
//   * the next tile's x window is requested from HBM before the tile's MFMAs start and written to LDS (LeakyReLU on the way) after them;
//   * an activation fragment (one ds_read_b128) feeds two MFMAs and no weight is ever read from LDS or L2 inside the loop.
// Output rows leave the accumulators as 16 bytes per lane: the A-operand rows are PERMUTED when the weights are fetched (row m of a
// wave's tile cc is output channel 8 (m >> 2) + 4 cc + (m & 3) of its 32), so that a lane's two 16x16 tiles hold 8 consecutive
// channels of one frame — one ds_write_b128 / global store per frame tile and lane, conflict-free at the 160-byte row stride the
// fragment reads want (convwin.hip's uint2 rows collided four ways: 23 % of that kernel's LDS cycles).
// Same fp16 / bf16 roundings, same MFMA shape and the same (tap, k-step) accumulation order as ttsk_hifi_conv_pair: bit-identical.
#include <type_traits>
#include "common.h"
#include "gemm_common.h"      // dma16: buffer_load_dwordx4 ... lds from inline asm

namespace {

struct WsArgs {
  const bf16_t* x;      // (B, len, 64) 16-bit, raw block input
  const bf16_t* w1;     // fragment-major packs [K][2][4][64][8] (ttsk_pack_resblock_weight)
  const bf16_t* w2;
  const float* b1;
  const float* b2;
  bf16_t* out;          // (B, len, 64)
  int len, tiles_per_utt, n_tiles, total_rows;
  int mode;             // 0: out = y   1: out += y   2: out = lrelu((out + y) * scale, final_slope)   (convwin.hip's modes)
  float slope, scale, final_slope;
#ifdef TTSK_STAMPS
  unsigned long long* stamps;   // diagnostic build only (make stamps; ttsk_hifi_conv_pair_ws_set_stamps): [workgroup][step < 16][wave 8][4] x s_memrealtime
#endif
};
#ifdef TTSK_STAMPS
#define WS_TSTAMP(k)                                                                                                              \
  do {                                                                                                                            \
    if (a.stamps && lane == 0 && s < 16) a.stamps[(int64_t)gridDim.x * 16 * 8 * 4 + (((int64_t)blockIdx.x * 16 + s) * 8 + wave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define WS_STAMP(k)                                                                                                               \
  do {                                                                                                                            \
    if (a.stamps && lane == 0 && s < 16) a.stamps[(((int64_t)blockIdx.x * 16 + s) * 8 + wave) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define WS_STAMP(k) do {} while (0)      // the product library carries no stamp code
#define WS_TSTAMP(k) do {} while (0)
#endif

// Geometry by channel count.  C = 64: two 32-channel groups x two frame halves per role, 192-frame tiles (c1 computes 13 frame tiles of 16, c2 12).
// C = 128 (k = 3 only: 96 weight registers per wave; k = 7 would need 224): four channel groups x all frames, 96-frame tiles (7 / 6 frame tiles:
// two windows of 288-byte rows per tile side by side must fit the 160 KiB).
template <int C, int K, int D>
struct WsGeom {
  static constexpr int HK = (K - 1) / 2;
  static constexpr int RS = C * 2 + 32;                 // row stride = 2 mod 4 sixteen-byte units: conflict-free fragment reads at every tap shift
  static constexpr int NCG = C / 32, NFG = 4 / NCG;     // output-channel groups of 32, frame groups: NCG * NFG = the role's four waves
  static constexpr int KS = C / 32, NFR = K * KS;       // k-steps per tap, fragments per frame tile
  static constexpr int TT = C == 64 ? 192 : 96, NT2 = TT / 16, NT1 = NT2 + 1, R1 = NT1 * 16;
  static constexpr int T1A = (NT1 + NFG - 1) / NFG;     // c1 frame tiles of frame group 0 (the other takes the rest), T2 each for c2
  static constexpr int T2 = NT2 / NFG;
  static constexpr int XROWS = R1 + 2 * HK * D;         // x-window row r <-> frame t0 - HK - HK*D + r  (c1 row r reads rows r + g*D)
  static constexpr int NPIECE = (XROWS * RS + 1023) / 1024;   // the window as 1 KiB LDS-DMA pieces (one wave instruction each)
  static constexpr int XBYTES = NPIECE * 1024;          // (whole pieces: a piece's tail lanes must not land in the next window)
  static constexpr int TBYTES = R1 * RS;                // t-window row r <-> frame t0 - HK + r         (c2 row p reads rows p + g)
  static constexpr int SMEM = 2 * XBYTES + 2 * TBYTES;  // 150,528 B at C = 64, k = 11, d = 5; 136,192 B at C = 128, k = 3, d = 5
  static constexpr int PPW = (NPIECE + 3) / 4;          // pieces per c1 wave
  static constexpr int PF = K >= 11 ? 2 : (K >= 7 ? 3 : T2);   // residual / running-sum rows of the c2 waves: frame tiles requested ahead
  static constexpr int DR = NFR % 6 == 0 ? 6 : (NFR % 7 == 0 ? 7 : 4);   // activation fragments in flight ahead of their MFMAs (a divisor of NFR where there is one)
  static_assert(T2 * NFG == NT2 && SMEM <= 160 * 1024, "tile geometry");
};
constexpr int ws_tile_frames(int C) { return C == 64 ? 192 : 96; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int WS_PAD = 0x7F000000;                     // a buffer offset past every utterance (< 2^30 bytes each): such a load returns 0, such a store is dropped

template <int C, int K, int D, int MODE, bool F16>
__global__ __launch_bounds__(512, 2) void pair_ws_kernel(const WsArgs a) {
  using G = WsGeom<C, K, D>;
  constexpr int HK = G::HK, RS = G::RS, XROWS = G::XROWS, NPIECE = G::NPIECE, PPW = G::PPW, DR = G::DR, NFR = G::NFR, KS = G::KS, NCG = G::NCG,
                TT = G::TT, NT1 = G::NT1, R1 = G::R1, T1A = G::T1A, T2 = G::T2;
  constexpr int PF = (K >= 11 && MODE != 0) ? 1 : G::PF;      // (k = 11 with the running sum's rows as well: one frame tile ahead is all the registers allow, and a tile is ~1 us there)
  constexpr bool PRE = K < 11;        // the window pieces' per-lane offsets kept in registers (k = 11: recomputed per tile, the weights need the room)
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: M0, scalar branches)
  const int l15 = lane & 15, q = lane >> 4;
  const int role = wave >> 2, ch = (wave & 3) % NCG, fh = (wave & 3) / NCG;      // channel group, frame group; waves w and w + 4 share a SIMD: same (ch, fh), c1 | c2
  const int len = a.len;
  const int tau0 = (int)((int64_t)blockIdx.x * a.n_tiles / gridDim.x), tau1 = (int)((int64_t)(blockIdx.x + 1) * a.n_tiles / gridDim.x);
  const int n = tau1 - tau0;
  // Every global access is a buffer access over ONE UTTERANCE (base = its first frame, length = len rows): a frame before or behind the
  // utterance is an offset outside the buffer — the load returns 0 (the convs' zero padding), the store is dropped — with no compare, no
  // select and no branch.  Branches would cut the tile loops into basic blocks the instruction scheduler cannot move MFMAs and LDS reads
  // across, and the address arithmetic of a masked access costs as many VALU slots as the MFMAs leave free.
  const unsigned ubytes = (unsigned)len * (C * 2);
  auto utt_rsrc = [&](const bf16_t* base, int bi, unsigned bytes) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base) + (int64_t)bi * len * C, 0, bytes, 0x00020000);
  };

  // ---- this wave's weights: K taps x KS k-steps x 2 output-channel tiles, rows permuted (see the header)
  bf16x8 w[K][KS][2];
  f32x4 bv[2];
  {
    const unsigned char* wp = (const unsigned char*)(role == 0 ? a.w1 : a.w2);
    const float* bp = role == 0 ? a.b1 : a.b2;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int co = 32 * ch + 8 * (l15 >> 2) + 4 * cc + (l15 & 3);
      const int off = (co >> 4) * 1024 + ((co & 15) + 16 * q) * 16;
#pragma unroll
      for (int g = 0; g < K; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[g][ks][cc] = *(const bf16x8*)(wp + (g * KS + ks) * (C / 16) * 1024 + off);
      bv[cc] = *(const f32x4*)(bp + 32 * ch + 8 * q + 4 * cc);
    }
  }
  // The weights have arrived before anything else starts, and the compiler knows it: left to its own bookkeeping it guards their first
  // uses INSIDE the tile loops with counted s_waitcnt vmcnt(N), which there wait for the loads this kernel keeps in flight on purpose.
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
  auto tile_pos = [&](int tau, int& bi, int& t0) __attribute__((always_inline)) {
    bi = tau / a.tiles_per_utt;
    t0 = (tau - bi * a.tiles_per_utt) * TT;
  };

  // ---- x window of tile tau: raw x straight into LDS by LDS-DMA (no registers: the request is a whole step ahead of its use), then
  //      LeakyReLU in place by the wave that requested the piece (no other wave's data: no barrier between the two).  A piece is 1 KiB of
  //      the window image (6.4 rows of 160 bytes at C = 64): lane -> (row, 16-byte column) by a division by RS / 16; the last two columns are the row's pad
  //      and rows past the window are the piece's tail: both requested outside the buffer (zeros).
  const int cwave = wave & 3;
  auto piece_off = [&](int j) __attribute__((always_inline)) {      // byte offset of this lane's 16 bytes of piece j * 4 + cwave from the window's first frame
    const int u = (j * 4 + cwave) * 64 + lane;
    const int row = u / (RS / 16), c = u - row * (RS / 16);
    return (c < C / 8 && row < XROWS) ? row * (C * 2) + c * 16 : WS_PAD;
  };
  int poff[PRE ? PPW : 1];
  if (PRE && role == 0) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) poff[j] = piece_off(j);
  }
  auto win_request = [&](int tau, unsigned xw_lds) __attribute__((always_inline)) {
    int bi, t0;
    tile_pos(tau, bi, t0);
    const __amdgpu_buffer_rsrc_t rxu = utt_rsrc(a.x, bi, ubytes);
    const int lo = (t0 - HK - HK * D) * (C * 2);
#pragma unroll
    for (int j = 0; j < PPW; ++j)
      if (j * 4 + cwave < NPIECE) dma16(rxu, xw_lds + (j * 4 + cwave) * 1024, (PRE ? poff[j] : piece_off(j)) + lo);
  };
  auto win_activate = [&](unsigned char* XWb) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed (asm requests are invisible to the compiler's counters)
    constexpr int CHK = K >= 11 ? 4 : PPW;      // pieces in flight through registers (k = 11: the weights leave room for four)
#pragma unroll
    for (int j0 = 0; j0 < PPW; j0 += CHK) {
      uint4 v[CHK];
#pragma unroll
      for (int j = j0; j < j0 + CHK && j < PPW; ++j)
        if (j * 4 + cwave < NPIECE) v[j - j0] = *(const uint4*)(XWb + (j * 4 + cwave) * 1024 + lane * 16);
#pragma unroll
      for (int j = j0; j < j0 + CHK && j < PPW; ++j)
        if (j * 4 + cwave < NPIECE)
          *(uint4*)(XWb + (j * 4 + cwave) * 1024 + lane * 16) = make_uint4(lrelu2_fast<F16>(v[j - j0].x, a.slope), lrelu2_fast<F16>(v[j - j0].y, a.slope),
                                                                           lrelu2_fast<F16>(v[j - j0].z, a.slope), lrelu2_fast<F16>(v[j - j0].w, a.slope));
    }
  };

  // ---- one frame tile (16 rows x this wave's 32 channels) of a conv: fragment f = (tap f / KS, k-step f % KS), DR fragments in flight;
  //      on return the ring holds the first DR fragments at `nxt`.  `dd`: the conv's dilation as a compile-time constant, so that
  //      every fragment distance is an instruction offset
  bf16x8 ring[DR];
  // The ring is indexed modulo DR and a tile has NFR = 2 K fragments: when DR does not divide NFR the next tile's fragment p does not land in
  // slot p but in slot (p + ROT') % DR, ROT' = (ROT + NFR) % DR — `rotc` carries that rotation from tile to tile as a compile-time constant
  // (k = 11: DR = 4, NFR = 22, the rotation alternates 0, 2, so the tile loops below advance two tiles per trip; the other instances have DR | NFR).
  constexpr int ROT1 = NFR % DR;
  static_assert((2 * NFR) % DR == 0, "the rotation must return to 0 after two tiles");
  auto conv_tile = [&](auto dd, auto rotc, f32x4 (&acc)[2], const unsigned char* base, const unsigned char* nxt) __attribute__((always_inline)) {
    constexpr int DD = decltype(dd)::value, ROT = decltype(rotc)::value;
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      const bf16x8 Bf = ring[(f + ROT) % DR];
      acc[0] = mfma16<F16>(w[f / KS][f % KS][0], Bf, acc[0]);
      acc[1] = mfma16<F16>(w[f / KS][f % KS][1], Bf, acc[1]);
      const int nf = f + DR;
      ring[(f + ROT) % DR] = nf < NFR ? *(const bf16x8*)(base + (nf / KS) * DD * RS + (nf % KS) * 64)
                                      : *(const bf16x8*)(nxt + ((nf - NFR) / KS) * DD * RS + ((nf - NFR) % KS) * 64);
    }
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
    }
  };
  auto ring_prime = [&](auto dd, const unsigned char* base) __attribute__((always_inline)) {
    constexpr int DD = decltype(dd)::value;
#pragma unroll
    for (int j = 0; j < DR; ++j) ring[j] = *(const bf16x8*)(base + (j / KS) * DD * RS + (j % KS) * 64);
    __builtin_amdgcn_sched_barrier(0);      // all of them in flight before the first MFMA, and not mistaken for the tile's own reads by its groups
  };
  using Rot0 = std::integral_constant<int, 0>;
  using Rot1 = std::integral_constant<int, ROT1>;
  using DilC1 = std::integral_constant<int, D>;
  using DilC2 = std::integral_constant<int, 1>;

  unsigned char* const XW0 = smem;
  unsigned char* const TW0 = smem + 2 * G::XBYTES;
  const unsigned xw0_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  if (role == 0) {        // prologue: the first tile's window
    win_request(tau0, xw0_lds);
    win_activate(XW0);
  }
  __syncthreads();

  // c2 waves: this lane's 8 channels of frame tile i0c + j of a tile; the requests of its residual rows (raw x) and, in the accumulating modes,
  // of the running sum
  const int i0c = fh * T2, lc2 = l15 * (C * 2) + (32 * ch + 8 * q) * 2;
  u32x4 rres[PF], rout[MODE ? PF : 1];
  auto req = [&](int tau, int j, int slot) __attribute__((always_inline)) {
    int bi, t0;
    tile_pos(tau, bi, t0);
    const int off = lc2 + (t0 + (i0c + j) * 16) * (C * 2);      // (rows behind the utterance: outside the buffer)
    rres[slot] = __builtin_amdgcn_raw_buffer_load_b128(utt_rsrc(a.x, bi, ubytes), off, 0, 0);
    if (MODE) rout[slot] = __builtin_amdgcn_raw_buffer_load_b128(utt_rsrc(a.out, bi, ubytes), off, 0, 0);
  };

#pragma unroll 1
  for (int s = 0; s <= n; ++s) {
    WS_STAMP(0);
    if (role == 0) {
      // ================= c1 waves: t = lrelu(c1(lrelu(x)) + b1) of tile s -> TW[s & 1]; x window of tile s + 1 -> XW[(s + 1) & 1]
      const bool more = s + 1 < n;
      unsigned char* XWn = XW0 + ((s + 1) & 1) * G::XBYTES;
      if (more) win_request(tau0 + s + 1, xw0_lds + ((s + 1) & 1) * G::XBYTES);
      if (s < n) {
        int bi, t0;
        tile_pos(tau0 + s, bi, t0);
        const unsigned char* XWb = XW0 + (s & 1) * G::XBYTES + l15 * RS + q * 16;
        unsigned char* TWb = TW0 + (s & 1) * G::TBYTES + l15 * RS + ch * 64 + q * 16;
        const int i0 = fh ? T1A : 0, i1 = (fh || G::NFG == 1) ? NT1 : T1A;
        const bool edge = t0 - HK < 0 || t0 - HK + R1 > len;      // the tile touches an end of the utterance: t is zero outside it (c2's padding)
        WS_TSTAMP(0);
        ring_prime(DilC1{}, XWb + i0 * 16 * RS);
        auto c1_tile = [&](auto rotc, int i) __attribute__((always_inline)) {
          f32x4 acc[2];
          conv_tile(DilC1{}, rotc, acc, XWb + i * 16 * RS, XWb + (i + 1 < i1 ? i + 1 : i) * 16 * RS);
          unsigned o[4];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            f32x4 v = acc[cc] + bv[cc];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
            o[cc * 2] = pack2<F16>(v[0], v[1]);
            o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
          }
          if (edge) {
            const int t = t0 - HK + i * 16 + l15;
            if (t < 0 || t >= len) o[0] = o[1] = o[2] = o[3] = 0u;
          }
          *(uint4*)(TWb + i * 16 * RS) = make_uint4(o[0], o[1], o[2], o[3]);
          WS_TSTAMP(1 + i - i0);
        };
#pragma unroll 1
        for (int i = i0; i + 1 < i1; i += 2) {
          c1_tile(Rot0{}, i);
          c1_tile(Rot1{}, i + 1);
        }
        if ((i1 - i0) & 1) c1_tile(Rot0{}, i1 - 1);
      }
      WS_STAMP(1);
      if (more) win_activate(XWn);
    } else {
      // ================= c2 waves: out = c2(t) + b2 + x (+ the MRF modes) of tile s - 1 from TW[(s - 1) & 1].  The residual rows (raw x) and
      // the running sum of a frame tile are requested PF frame tiles ahead, ACROSS the step boundary: the first PF of tile s while tile
      // s - 1 is being finished (at step 0: while the c1 waves produce tile 0), so that no step starts by waiting for L2.
      if (s == 0) {
#pragma unroll
        for (int j = 0; j < PF; ++j) req(tau0, j, j);
      } else {
        int bi, t0;
        tile_pos(tau0 + s - 1, bi, t0);
        const __amdgpu_buffer_rsrc_t rou = utt_rsrc(a.out, bi, ubytes);
        const int off0 = lc2 + (t0 + i0c * 16) * (C * 2);
        const unsigned char* TWb = TW0 + ((s - 1) & 1) * G::TBYTES + l15 * RS + q * 16;
        WS_TSTAMP(0);
        ring_prime(DilC2{}, TWb + i0c * 16 * RS);
#pragma unroll
        for (int j = 0; j < T2; ++j) {
          const int i = i0c + j;
          f32x4 acc[2];
          if (j & 1) conv_tile(DilC2{}, Rot1{}, acc, TWb + i * 16 * RS, TWb + (j + 1 < T2 ? i + 1 : i) * 16 * RS);
          else conv_tile(DilC2{}, Rot0{}, acc, TWb + i * 16 * RS, TWb + (j + 1 < T2 ? i + 1 : i) * 16 * RS);
          u32x4 o;
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            f32x4 v = acc[cc] + bv[cc];
            float r0, r1, r2, r3;
            unpack2<F16>(rres[j % PF][cc * 2], r0, r1); unpack2<F16>(rres[j % PF][cc * 2 + 1], r2, r3);
            v += f32x4{r0, r1, r2, r3};
            if constexpr (MODE != 0) {
              unpack2<F16>(rout[j % PF][cc * 2], r0, r1); unpack2<F16>(rout[j % PF][cc * 2 + 1], r2, r3);
              v += f32x4{r0, r1, r2, r3};
            }
            if constexpr (MODE == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] *= a.scale; v[e] = fmaxf(v[e], v[e] * a.final_slope); }
            }
            o[cc * 2] = pack2<F16>(v[0], v[1]);
            o[cc * 2 + 1] = pack2<F16>(v[2], v[3]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(o, rou, off0 + j * 16 * (C * 2), 0, 0);      // (rows behind the utterance: dropped)
          WS_TSTAMP(1 + j);
          if (j + PF < T2) req(tau0 + s - 1, j + PF, j % PF);
          else req(s < n ? tau0 + s : tau0 + s - 1, j + PF - T2, j % PF);      // (behind the last tile: a harmless repeat, so that the request count stays fixed)
        }
      }
      WS_STAMP(1);
    }
    WS_STAMP(2);
    __syncthreads();
    WS_STAMP(3);
  }
}

template <int C, int K, int D>
void launch_ws(const WsArgs& a, int f16, int grid, hipStream_t s) {
  dim3 g(grid), b(512);
  if (f16) {
    if (a.mode == 0) hipLaunchKernelGGL((pair_ws_kernel<C, K, D, 0, true>), g, b, 0, s, a);
    else if (a.mode == 1) hipLaunchKernelGGL((pair_ws_kernel<C, K, D, 1, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((pair_ws_kernel<C, K, D, 2, true>), g, b, 0, s, a);
  } else {
    if (a.mode == 0) hipLaunchKernelGGL((pair_ws_kernel<C, K, D, 0, false>), g, b, 0, s, a);
    else if (a.mode == 1) hipLaunchKernelGGL((pair_ws_kernel<C, K, D, 1, false>), g, b, 0, s, a);
    else hipLaunchKernelGGL((pair_ws_kernel<C, K, D, 2, false>), g, b, 0, s, a);
  }
}

}  // namespace

#ifdef TTSK_STAMPS
static unsigned long long* g_ws_stamps = nullptr;
// diagnostic build only (`make stamps`, tools/debug/pairws_stamps.py; not declared in ttsk.h, not in the product library)
extern "C" int ttsk_hifi_conv_pair_ws_set_stamps(void* dev_buffer) {
  g_ws_stamps = (unsigned long long*)dev_buffer;
  return TTSK_OK;
}
#endif

extern "C" int ttsk_hifi_conv_pair_ws_supported(int C, int K, int dil) {
  return ((C == 64 && (K == 3 || K == 7 || K == 11)) || (C == 128 && K == 3)) && (dil == 1 || dil == 3 || dil == 5);
}

extern "C" int ttsk_hifi_conv_pair_ws(const void* x16, const void* w1_pack, const float* bias1, const void* w2_pack, const float* bias2,
                                      void* out16, int f16, int B, int len, int C, int K, int dil, float slope, int mode, float scale,
                                      float final_slope, int max_wgs, void* stream) {
  TTSK_REQUIRE(x16 && w1_pack && bias1 && w2_pack && bias2 && out16, "ttsk_hifi_conv_pair_ws: null pointer");
  TTSK_REQUIRE(B > 0 && len > 0 && B <= 65535 && x16 != out16, "ttsk_hifi_conv_pair_ws: bad sizes / in-place output");
  TTSK_REQUIRE(ttsk_hifi_conv_pair_ws_supported(C, K, dil), "ttsk_hifi_conv_pair_ws: no instance for C=%d K=%d dil=%d", C, K, dil);
  TTSK_REQUIRE(mode >= 0 && mode <= 2 && (final_slope > 0.f || mode != 2), "ttsk_hifi_conv_pair_ws: bad mode / final_slope");
  TTSK_REQUIRE(slope > 0.f && slope < 1.f, "ttsk_hifi_conv_pair_ws: LeakyReLU slope %g outside (0, 1)", slope);
  TTSK_REQUIRE(((((uintptr_t)x16) | ((uintptr_t)w1_pack) | ((uintptr_t)w2_pack) | ((uintptr_t)bias1) | ((uintptr_t)bias2) | ((uintptr_t)out16)) & 15) == 0,
               "ttsk_hifi_conv_pair_ws: 16-byte alignment");
  const int tt = ws_tile_frames(C);
  const int tiles_per_utt = (len + tt - 1) / tt;
  const int64_t n_tiles = (int64_t)B * tiles_per_utt;
  TTSK_REQUIRE((int64_t)len * (C * 2) < (1ll << 30), "ttsk_hifi_conv_pair_ws: utterances of 1 GiB or more (32-bit buffer offsets)");
  WsArgs a{(const bf16_t*)x16, (const bf16_t*)w1_pack, (const bf16_t*)w2_pack, bias1, bias2, (bf16_t*)out16, len, tiles_per_utt, (int)n_tiles,
           B * len, mode, slope, scale, final_slope};
#ifdef TTSK_STAMPS
  a.stamps = g_ws_stamps;
#endif
  int grid = max_wgs > 0 ? max_wgs : 256;        // one persistent workgroup per CU
  if (grid > n_tiles) grid = (int)n_tiles;
  hipStream_t s = (hipStream_t)stream;
  switch ((C == 128 ? 1000 : 0) + K * 10 + dil) {
    case 31: launch_ws<64, 3, 1>(a, f16, grid, s); break;
    case 33: launch_ws<64, 3, 3>(a, f16, grid, s); break;
    case 35: launch_ws<64, 3, 5>(a, f16, grid, s); break;
    case 71: launch_ws<64, 7, 1>(a, f16, grid, s); break;
    case 73: launch_ws<64, 7, 3>(a, f16, grid, s); break;
    case 75: launch_ws<64, 7, 5>(a, f16, grid, s); break;
    case 111: launch_ws<64, 11, 1>(a, f16, grid, s); break;
    case 113: launch_ws<64, 11, 3>(a, f16, grid, s); break;
    case 115: launch_ws<64, 11, 5>(a, f16, grid, s); break;
    case 1031: launch_ws<128, 3, 1>(a, f16, grid, s); break;
    case 1033: launch_ws<128, 3, 3>(a, f16, grid, s); break;
    default: launch_ws<128, 3, 5>(a, f16, grid, s); break;
  }
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
