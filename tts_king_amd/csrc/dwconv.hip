// dwconv.hip — weight gradient of a Conv1d with taps (the FFT blocks' w_1, k = 9: 62.6 % of the step's FLOPs with its forward and
// input gradient), as a kernel of its own instead of `taps` batched problems of the grouped GEMM.
// reference: fs_two/transformer/SubLayers.py:93-101 (w_1 = Conv1d(256, 1024, k = 9)); what torch's conv backward computes for
// `weight.grad`:  dW[co][tap][ci] = sum_b sum_t dY[b, t, co] * X[b, t + tap - k/2, ci]   (rows outside the utterance are the conv's
// zero padding).
//
// Why: counters on gemm2_group_kernel<256, true, true> alone on the chip (profiles/r03_pmc_dw.txt) say it is bound by operand
// delivery, not by LDS reads or the MFMA pipe: MFMA busy 31 %, LDS array 17 %, waves parked on s_waitcnt / the K-step barrier 35 % of
// their cycles; it pulls 48 KiB per K step per CU through L2 -> LDS (4.3 GB per step's weight gradients, 71 % L2 hits) and runs at the
// 33 GB/s per CU that path gives.  Per tap it re-fetches the whole dY tile (256 columns) and the X tile (128 columns), 18 times per
// (Cout, Cin) pair of w_1.
// Here a wave owns 64 output channels x ALL taps x a 32-column slice of the input channels: the dY rows are fetched once for all taps
// and the X rows as ONE window (32 + k - 1 rows) that every tap reads at its own row offset:
//   k = 9: 6.5 KiB of LDS-DMA per wave and K step (32 rows) for 72 MFMAs (16x16x32) instead of 6 KiB per 32 — 2.1x fewer bytes per FLOP.
// The four waves of a workgroup share nothing: each stages its own operands (no barrier anywhere in the K loop); 512 registers per wave
// (288 accumulators), one workgroup per CU.  Fills go global -> registers -> LDS (buffer_load_dwordx4 + ds_write_b128), two K steps of
// loads in flight in two register sets, the LDS double-buffered: with LDS-DMA (buffer_load ... lds), which the first build used, every
// 1 KiB piece cost the issuing wave ~130 cycles of issue time — 8 pieces per 72 MFMAs, one wave per SIMD and nobody to fill the
// MFMA pipe meanwhile: 261 us for the six decoder weights against 177 us with the fills switched off (tools/debug/dwconv_micro.py).
// K steps follow the utterances: 32 rows of ONE utterance per step, rows past the utterance's own length (lens, when given: PAD rows
// carry no gradient, Layers.py:29,32) are never fetched — the row count drops from B*S to sum(ceil(len_b / 32) * 32).
//
// LDS images are laid out so that every fragment read is ONE lane-constant base register plus an immediate (the first version, with an
// XOR swizzle that depended on the tap's row offset, kept 90 address registers live and spilled 111):
//   * the contraction index of MFMA slot (lane group lg = lane >> 4, element e = 0..7) is row 16*(lg>>1) + 8*(e>>2) + 4*(lg&1) + (e&3) of
//     the step's 32 rows — the same permutation for both operands, so the sum is unchanged — which makes the 32 lanes that one
//     ds_read_b64_tr_b16 cycle serves touch 8 CONSECUTIVE rows;
//   * X window: one image per 16-channel group, [64 rows][32 B]: 8 consecutive rows = 256 contiguous bytes = all 64 banks, for any first
//     row, so a tap is an immediate offset of tap*32 bytes;
//   * dY rows: [32 rows][128 B] as four 1 KiB pieces (8 rows each, fetched as full 128-byte row segments); the 16-channel block i of row
//     r sits at block i ^ ((r & 7) >> 1) (applied on the DMA's source side): the 8 rows of one read are one piece, conflict-free.
#include "gemm_common.h"

namespace {

constexpr int DWC_NW = 4, DWC_NT = DWC_NW * 64, DWC_MAXP = 12, DWC_OOB = 0x7FFFFFFF;

struct DwcProblem {
  const bf16_t* dy;         // [B*S][ldy], columns [0, Cout)
  const bf16_t* x;          // [B*S][ldx], columns [0, Cin)
  float* dw;                // [Cout][TAPS][Cin]
  const long long* lens;    // [B] rows of each utterance that carry a gradient, or null (all S)
  int Cout, Cin, ldy, ldx, B, S, accumulate;
  int wg0, nwg, tiles_ci;   // workgroups [wg0, wg0 + nwg) of the launch, in XCD-major order
};
struct DwcArgs {
  DwcProblem p[DWC_MAXP];
  int n, total;
};

template <int TAPS, int CI16, int NSTAGE_>
struct DwcCfg {
  static constexpr int BK = 32;                                 // rows per K step
  static constexpr int CPAD = TAPS / 2;
  static constexpr int W = BK + 2 * CPAD;                       // window rows of X per K step
  static constexpr int NPA = 4;                                 // pieces of the dY rows: 32 rows x 128 B
  static constexpr int NPB = 2 * CI16;                          // X window: per 16-channel group 64 image rows x 32 B = two pieces
  static constexpr int NP = NPA + NPB;
  static constexpr int A_BYTES = NPA * 1024, B_BYTES = NPB * 1024, STAGE = A_BYTES + B_BYTES;
  static constexpr int NSTAGE = NSTAGE_;                        // ring depth: NSTAGE - 1 K steps in flight
  static constexpr int NTILE = TAPS * CI16;                     // 16-column output tiles per wave: (tap, 16-channel group)
  static constexpr int WAVE_BYTES = NSTAGE * STAGE;
  static constexpr int SMEM = DWC_NW * WAVE_BYTES;
  static_assert(SMEM <= 163840, "LDS");
  static_assert(W <= 64, "window rows fit the two pieces of an image");
  static_assert(NSTAGE == 2, "LDS double buffer");
};

// The wave's accumulators are 288 registers (k = 9): more than the 256 AGPRs hipcc gives a kernel that uses AGPRs at all (it splits the
// 512-entry file 256 / 256), and left to itself it shuttles the overflow between the two classes (2,356 v_accvgpr moves, 263 spills in
// the first build).  The MFMA is therefore issued from inline asm with the accumulator's register class spelled out: column tiles
// 0 .. NTILE_A-1 live in AGPRs, the rest in VGPRs (32 of them at k = 9), and no accumulator ever moves.  No MFMA here reads an
// accumulator another MFMA wrote less than a whole K step (72 MFMAs) earlier, so no hazard nops are needed; the operands come from
// ds_reads, whose waits the compiler still inserts for asm operands.
template <bool IN_AGPR>
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8 a, const bf16x8 b) {
  if constexpr (IN_AGPR) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int TAPS, int CI16, int NSTAGE_>
__global__ __launch_bounds__(DWC_NT, 1) void dwconv_kernel(const DwcArgs args) {
  using CF = DwcCfg<TAPS, CI16, NSTAGE_>;
  constexpr int BK = CF::BK, CPAD = CF::CPAD, W = CF::W, NPA = CF::NPA, NP = CF::NP;
  constexpr int A_BYTES = CF::A_BYTES, STAGE = CF::STAGE, NSTAGE = CF::NSTAGE, NTILE = CF::NTILE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[CF::SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- which tile: XCD-major workgroup id (consecutive ids sit on ONE XCD: the input-channel slices that share a dY tile, and the
  // output-channel tiles that share an X slice, meet in one L2)
  int id;
  {
    const int G = gridDim.x, wg = blockIdx.x, q = G >> 3, r = G & 7, x = wg & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
  }
  int pi = 0;
  for (int i = 1; i < args.n; ++i)
    if (id >= args.p[i].wg0) pi = i;
  pi = __builtin_amdgcn_readfirstlane(pi);
  const DwcProblem& P = args.p[pi];
  const int local = id - P.wg0;
  if (local >= P.nwg) return;
  const int tile_ci = local % P.tiles_ci, tile_m = local / P.tiles_ci;
  const int co0 = tile_m * 256 + wave * 64, ci0 = tile_ci * (16 * CI16);
  const int S = P.S, nB = P.B, ldy = P.ldy, ldx = P.ldx;

  // rows of every utterance that are walked, one per lane (B <= 64), and the number of K steps
  int nb_lane = 0;
  if (lane < nB) {
    nb_lane = S;
    if (P.lens) { const long long v = P.lens[lane]; nb_lane = v < 0 ? 0 : (v > S ? S : (int)v); }
  }
  int nsteps = 0;
  for (int b = 0; b < nB; ++b) nsteps += (__builtin_amdgcn_readlane(nb_lane, b) + BK - 1) / BK;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.dy, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, 0x7FFFFFF0, 0x00020000);
  unsigned char* wsm = smem + wave * CF::WAVE_BYTES;

  // ---- per-lane source coordinates of the LDS-DMA pieces (a piece = 1 KiB of the LDS image, lane-linear: any layout trick goes on the
  // source side).  dY piece q = rows q*8 .. +7: lane -> row lane/8, 16-byte chunk lane%8, fetching block (chunk/2) ^ (row/2 & 3).
  const int a_kl = lane >> 3, a_c = lane & 7;
  const int a_base = (a_kl * ldy + co0 + (((a_c >> 1) ^ ((a_kl >> 1) & 3)) * 16) + (a_c & 1) * 8) * 2;
  // X piece (cit, half) = image rows half*32 .. +31 of channel group cit: lane -> row lane/2, 16-byte chunk lane%2
  const int b_wl = lane >> 1, b_c = lane & 1;
  const int b_base = (b_wl * ldx + ci0 + b_c * 8) * 2;

  // ---- load cursor: the next K step to fetch = rows [ij*32, ij*32 + 32) of utterance ib
  int ib = 0, ij = 0, inb = 0, loaded = 0;
  while (ib < nB && (inb = __builtin_amdgcn_readlane(nb_lane, ib)) == 0) ++ib;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  auto load_piece = [&](int q, bool live) __attribute__((always_inline)) -> u32x4 {
    // q = 0..NPA-1: dY rows; NPA..NP-1: X window.  Rows that do not exist (past the utterance's walked length for dY; outside
    // [0, S) for the window = the conv's zero padding; past the window; `live` false = past the last K step) are out-of-range offsets:
    // the hardware returns zeros without touching memory — no branch around a load anywhere in the loop.
    if (q < NPA) {
      const bool ok = live && ij * BK + q * 8 + a_kl < inb;
      const int off = a_base + ((ib * S + ij * BK + q * 8) * ldy) * 2;
      return __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? off : DWC_OOB, 0, 0);
    } else {
      const int qb = q - NPA, cit = qb >> 1, half = qb & 1;
      const int w = half * 32 + b_wl;
      const int t = ij * BK - CPAD + w;
      const bool ok = live && t >= 0 && t < S && w < W;
      const int off = b_base + ((ib * S + ij * BK - CPAD + half * 32) * ldx + cit * 16) * 2;
      return __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? off : DWC_OOB, 0, 0);
    }
  };
  auto load_advance = [&]() __attribute__((always_inline)) {
    ++loaded;
    if (++ij * BK >= inb) {
      ij = 0;
      ++ib;
      while (ib < nB && (inb = __builtin_amdgcn_readlane(nb_lane, ib)) == 0) ++ib;
    }
  };
  // piece q of a stage lands lane-linear, like an LDS-DMA piece: 16 bytes per lane at q*1024 + lane*16
  auto store_piece = [&](unsigned char* stage, int q, u32x4 v) __attribute__((always_inline)) {
    *(u32x4*)(stage + q * 1024 + lane * 16) = v;
  };

  f32x4 acc[4][NTILE];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTILE; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment reads.  Lane (lg, l15) supplies the 8-byte chunk (l15 & 3) of row r0 + 8*h (h = 0 / 1: the low / high four contraction
  // slots), r0 = 16*(lg>>1) + 4*(lg&1) + (l15>>2); after the transpose it holds, for output row / column l15, the slots 8*lg .. 8*lg+7.
  const int r0 = 16 * (lg >> 1) + 4 * (lg & 1) + (l15 >> 2), cq = (l15 & 3) << 3;
  int a_lane[4];               // dY: piece 2*(lg>>1) (+ h), row r0 & 7 inside it, block i ^ ((r0 & 7) >> 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) a_lane[i] = 2 * (lg >> 1) * 1024 + (r0 & 7) * 128 + ((i ^ ((r0 & 7) >> 1)) << 5) + cq;
  const int b_lane = A_BYTES + r0 * 32 + cq;     // X: image row r0 (+ 8*h + tap), 32 bytes per row
  typedef __attribute__((address_space(3))) bf16x4* tr_ptr;
  auto read_a = [&](const unsigned char* st, int i) __attribute__((always_inline)) -> bf16x8 {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + a_lane[i]));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + a_lane[i] + 1024));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto read_b = [&](const unsigned char* st, int j) __attribute__((always_inline)) -> bf16x8 {
    const int tap = j / CI16, cit = j % CI16;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + b_lane + cit * 2048 + tap * 32));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + b_lane + cit * 2048 + (tap + 8) * 32));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

#define DWC_SB() __builtin_amdgcn_sched_barrier(0)
  if (nsteps > 0) {
    // Register sets: set[n & 1] carries the pieces of K step n from their loads (issued during step n - 3) to their LDS stores (during
    // step n - 1); step n reads LDS stage n & 1.  So step s touches ONE set, set[(s + 1) & 1]: piece q is stored (for step s + 1) and the
    // register then takes the load of piece q of step s + 3; the other set's loads (step s + 2) stay in flight.
    u32x4 set[2][NP];
    unsigned char* stg[2] = {wsm, wsm + STAGE};
    const int nsteps2 = (nsteps + 1) & ~1;
    // prologue: steps 0 and 1 loaded, step 0 stored, step 2 loaded behind it
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int q = 0; q < NP; ++q) set[n][q] = load_piece(q, loaded < nsteps);
      if (loaded < nsteps) load_advance();
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) store_piece(stg[0], q, set[0][q]);
    {
      const bool more = loaded < nsteps;
#pragma unroll
      for (int q = 0; q < NP; ++q) set[0][q] = load_piece(q, more);
      if (more) load_advance();
    }
    bf16x8 af[2][4], bf[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = read_a(stg[0], i);
    bf[0] = read_b(stg[0], 0);
    bf[1] = read_b(stg[0], 1);
    // One K step: 72 (k = 9) MFMAs; between them, one piece after every PIECE_EVERY column tiles: the store of the next step's piece and
    // the load that refills its registers; the first fragments of the NEXT step are read during the last tiles.
    constexpr int PIECE_EVERY = (NTILE - 6) / NP;
    constexpr int NTILE_A = NTILE < 12 ? NTILE : 12;          // column tiles whose accumulators are AGPRs; the others live in VGPRs
    static_assert(PIECE_EVERY >= 1 && NTILE % 3 == 0 && NTILE >= 8, "slot plan");
    auto do_step = [&](int step, int par, bf16x8 (&ac)[4], bf16x8 (&an)[4]) __attribute__((always_inline)) {
      // par = step & 1 (a constant in each of the two copies of the body): reads stage `par`, works on set / stage `par ^ 1`
      const bool more = loaded < nsteps;           // a step to load (three ahead); past the end the registers take zeros
      const unsigned char* st = stg[par];
      unsigned char* stn = stg[par ^ 1];
      DWC_SB();
#pragma unroll
      for (int sl = 0; sl < NTILE; ++sl) {
        if (sl + 2 < NTILE) bf[(sl + 2) % 3] = read_b(st, sl + 2);
        if (sl % PIECE_EVERY == 0 && sl / PIECE_EVERY < NP) {
          const int q = sl / PIECE_EVERY;
          store_piece(stn, q, set[par ^ 1][q]);        // (for the step after the last one: into a stage nobody reads)
          set[par ^ 1][q] = load_piece(q, more);
          if (q == NP - 1 && more) load_advance();
        }
        if (sl >= NTILE - 6 && sl < NTILE - 2) an[sl - (NTILE - 6)] = read_a(stn, sl - (NTILE - 6));
        if (sl >= NTILE - 2) bf[(sl + 2) % 3] = read_b(stn, sl + 2 - NTILE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (sl < NTILE_A) mfma_acc<true>(acc[i][sl], ac[i], bf[sl % 3]);      // (folds after unrolling: sl is a constant)
          else mfma_acc<false>(acc[i][sl], ac[i], bf[sl % 3]);
        }
        DWC_SB();
      }
    };
    // the body exists twice (the two register sets / LDS stages alternate); an odd number of steps is rounded up with a step of zeros
    // rather than a third copy of the body (whose different register assignment cost ~500 moves and the spills at the loop's exit)
#pragma unroll 1
    for (int step = 0; step < nsteps2; step += 2) {
      do_step(step, 0, af[0], af[1]);
      do_step(step + 1, 1, af[1], af[0]);
    }
  }
#undef DWC_SB

  // ---- epilogue: the wave's 64 x (TAPS x CI16*16) tile straight from the accumulators (lane: rows 4*lg .. +3, column l15 of each
  // 16 x 16 fragment: 16 lanes cover 64 contiguous bytes of a (co, tap) row).  Accumulating: the old values of one 16-row band are all
  // requested before the first is used (one round trip per band instead of one per element).
  float* __restrict__ dw = P.dw;
  const int Cin = P.Cin;
  if (P.accumulate) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* row = dw + ((int64_t)(co0 + i * 16 + 4 * lg + r) * TAPS) * Cin + ci0 + l15;
        float old[NTILE];
#pragma unroll
        for (int j = 0; j < NTILE; ++j) old[j] = row[(j / CI16) * Cin + (j % CI16) * 16];
#pragma unroll
        for (int j = 0; j < NTILE; ++j) row[(j / CI16) * Cin + (j % CI16) * 16] = old[j] + acc[i][j][r];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* row = dw + ((int64_t)(co0 + i * 16 + 4 * lg + r) * TAPS) * Cin + ci0 + l15;
#pragma unroll
        for (int j = 0; j < NTILE; ++j) row[(j / CI16) * Cin + (j % CI16) * 16] = acc[i][j][r];
      }
    }
  }
}

template <int TAPS, int CI16, int NSTAGE_>
int launch_dwconv(const ttsk_dwconv_item* items, int n, hipStream_t s) {
  DwcArgs a;
  a.n = n;
  int wg = 0;
  for (int i = 0; i < n; ++i) {
    const ttsk_dwconv_item& it = items[i];
    DwcProblem& p = a.p[i];
    p.dy = (const bf16_t*)it.dy; p.x = (const bf16_t*)it.x; p.dw = it.dw; p.lens = (const long long*)it.lens;
    p.Cout = it.Cout; p.Cin = it.Cin; p.ldy = it.ldy; p.ldx = it.ldx; p.B = it.B; p.S = it.S; p.accumulate = it.accumulate;
    p.tiles_ci = it.Cin / (16 * CI16);
    p.nwg = (it.Cout / 256) * p.tiles_ci;
    p.wg0 = wg;
    wg += p.nwg;
  }
  a.total = wg;
  hipLaunchKernelGGL((dwconv_kernel<TAPS, CI16, NSTAGE_>), dim3(wg), dim3(DWC_NT), 0, s, a);
  return 0;
}

}  // namespace

extern "C" int ttsk_dwconv_supported(int Cout, int Cin, int K) {
  return K == 9 && Cout > 0 && Cin > 0 && Cout % 256 == 0 && Cin % 32 == 0;
}

extern "C" int ttsk_dwconv_batch(const ttsk_dwconv_item* items, int n, void* stream) {
  TTSK_REQUIRE(items && n > 0 && n <= DWC_MAXP, "dwconv_batch: 1..%d items", DWC_MAXP);
  const int K = items[0].K;
  for (int i = 0; i < n; ++i) {
    const ttsk_dwconv_item& it = items[i];
    TTSK_REQUIRE(it.dy && it.x && it.dw, "dwconv_batch: null pointer in item %d", i);
    TTSK_REQUIRE(it.K == K, "dwconv_batch: one kernel size per launch (item %d has %d, item 0 has %d)", i, it.K, K);
    TTSK_REQUIRE(ttsk_dwconv_supported(it.Cout, it.Cin, it.K), "dwconv_batch: no instance for Cout=%d Cin=%d K=%d", it.Cout, it.Cin, it.K);
    TTSK_REQUIRE(it.B > 0 && it.B <= 64 && it.S > 0, "dwconv_batch: 1..64 utterances of S > 0 rows");
    TTSK_REQUIRE(it.ldy >= it.Cout && it.ldx >= it.Cin && (it.ldy & 7) == 0 && (it.ldx & 7) == 0, "dwconv_batch: row pitches (multiples of 8, >= channels)");
    TTSK_REQUIRE((int64_t)it.B * it.S * (it.ldy > it.ldx ? it.ldy : it.ldx) * 2 < 0x7FFFFFF0ll, "dwconv_batch: operand beyond the 2 GiB buffer range");
    TTSK_REQUIRE(((((uintptr_t)it.dy) | ((uintptr_t)it.x) | ((uintptr_t)it.dw)) & 15) == 0, "dwconv_batch: 16-byte alignment");
  }
  launch_dwconv<9, 2, 2>(items, n, (hipStream_t)stream);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
