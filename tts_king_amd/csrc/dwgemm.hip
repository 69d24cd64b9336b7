// dwgemm.hip — weight gradients that are plain contractions over the rows (a Linear's, a k = 1 conv's, or one tap of a conv's):
//   dW[co][tap][ci] (+)= sum_b sum_{t < len_b} dY[b, t, co] * X[b, t + tap - k/2, ci]
// for Cout and Cin multiples of 256: the FFT blocks' w_2, q|k|v and fc (SubLayers.py:41-43,62,97) and the PostNet's 512 -> 512 convs
// (Layers.py:85-129), 40 % of the step's weight-gradient FLOPs.  reference: what torch's linear / conv backward leaves in `weight.grad`.
//
// Why not gemm2_group_kernel<256, true, true>: alone on the chip it is bound by operand delivery (profiles/r03_pmc_dw.txt: MFMA busy
// 31 %, waves parked on s_waitcnt / barrier 35 %), and what it pays per delivered byte is the LDS-DMA issue cost (~130 cycles of the
// issuing wave per 1 KiB piece, measured on dwconv.hip's first build).  This kernel keeps that kernel's shape of work — a workgroup
// per output tile and K range, operands shared by the waves through LDS, one barrier per K step — with
//   * a 256 x 256 tile (8 waves as 4 x 2, each 64 x 128): 32 KiB of operands per 32-row K step for 256 MFMAs, 1.5x fewer bytes per FLOP;
//   * fills through registers (buffer_load_dwordx4 + ds_write_b128), two K steps of loads in flight per wave, three LDS stages, so that
//     the next step's first fragments are read before the step's barrier;
//   * dwconv.hip's LDS images (one [32 rows][32 B] image per 16-column tile, contraction slots permuted so that one transposing read
//     touches 8 consecutive rows): every fragment read is a lane-constant base plus an immediate;
//   * K steps that follow the utterances (32 rows of one utterance; rows past lens[b] skipped), K split by utterance ranges into fp32
//     slabs [split][tap][Cout][Cin] that ttsk_gemm_reduce_batch sums (splits = 1: straight into dW).
#include "gemm_common.h"

namespace {

constexpr int DWG_NW = 8, DWG_NT = DWG_NW * 64, DWG_MAXP = 28, DWG_OOB = 0x7FFFFFFF;
constexpr int DWG_BK = 32, DWG_IMG = 1024 + 32;          // image pitch: 32 bytes of padding rotate consecutive tiles over the banks
constexpr int DWG_OPND = 16 * DWG_IMG;                    // one operand of a stage: 16 tiles
constexpr int DWG_STAGE = 2 * DWG_OPND, DWG_NSTAGE = 3;
constexpr int DWG_SMEM = DWG_NSTAGE * DWG_STAGE;          // 101,376 B

struct DwgProblem {
  const bf16_t* dy;
  const bf16_t* x;
  float* out;               // dW [Cout][taps][Cin] (splits == 1) or slabs [splits][taps][Cout][Cin]
  const long long* lens;
  int Cout, Cin, taps, ldy, ldx, B, S, accumulate, splits;
  int wg0, nwg, tiles_m, tiles_n;
};
struct DwgArgs {
  DwgProblem p[DWG_MAXP];
  int n;
};

__device__ __forceinline__ void dwgemm_tile(const DwgArgs& args, int id, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  int pi = 0;
  for (int i = 1; i < args.n; ++i)
    if (id >= args.p[i].wg0) pi = i;
  pi = __builtin_amdgcn_readfirstlane(pi);
  const DwgProblem& P = args.p[pi];
  int local = id - P.wg0;
  const int tile_n = local % P.tiles_n; local /= P.tiles_n;
  const int tap = local % P.taps; local /= P.taps;
  const int tile_m = local % P.tiles_m;
  const int split = local / P.tiles_m;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int S = P.S, ldy = P.ldy, ldx = P.ldx;
  const int shift = tap - P.taps / 2;
  const int b0 = (int)((long long)split * P.B / P.splits), b1 = (int)((long long)(split + 1) * P.B / P.splits);
  const int nB = b1 - b0;                      // utterances b0 .. b1-1 (at most 64)

  int nb_lane = 0;
  if (lane < nB) {
    nb_lane = S;
    if (P.lens) { const long long v = P.lens[b0 + lane]; nb_lane = v < 0 ? 0 : (v > S ? S : (int)v); }
  }
  int nsteps = 0;
  for (int b = 0; b < nB; ++b) nsteps += (__builtin_amdgcn_readlane(nb_lane, b) + DWG_BK - 1) / DWG_BK;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.dy, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, 0x7FFFFFF0, 0x00020000);

  // ---- fills.  A K step = 32 rows x 256 columns of each operand = 16 pieces of 2 rows x 512 B; wave w moves pieces 2w, 2w+1 of both.
  // Lane -> row 2p + lane/32, 16-byte chunk lane%32 (8 columns): tile chunk/2, half chunk%2 -> image byte tile*IMG + row*32 + half*16.
  const int f_row = lane >> 5, f_c = lane & 31;
  const int f_lds = (f_c >> 1) * DWG_IMG + f_row * 32 + (f_c & 1) * 16;           // + p*64 (two rows per piece)
  const int a_col = (m0 + f_c * 8) * 2, b_col = (n0 + f_c * 8) * 2;
  int ib = 0, ij = 0, inb = 0, loaded = 0;
  while (ib < nB && (inb = __builtin_amdgcn_readlane(nb_lane, ib)) == 0) ++ib;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  auto load_piece = [&](int q, bool live) __attribute__((always_inline)) -> u32x4 {
    // q = 0, 1: dY pieces 2*wave + q; q = 2, 3: X pieces.  Rows past the utterance's walked length / outside [0, S) under the tap's
    // shift / past the last K step: out-of-range offsets, zeros without a memory access.
    const int p = 2 * wave + (q & 1);
    const int row = ij * DWG_BK + 2 * p + f_row;
    // (offsets computed unconditionally, then ONE select: with the address arithmetic inside the conditional hipcc built branches
    // around it and drained vmcnt(0) at their join — every K step waited for the loads it had just issued)
    if (q < 2) {
      const int off = ((b0 + ib) * S + row) * ldy * 2 + a_col;
      const bool ok = live & (row < inb);
      return __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? off : DWG_OOB, 0, 0);
    } else {
      const int t = row + shift;
      const int off = ((b0 + ib) * S + t) * ldx * 2 + b_col;
      const bool ok = live & (row < inb) & (t >= 0) & (t < S);
      return __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? off : DWG_OOB, 0, 0);
    }
  };
  auto load_advance = [&]() __attribute__((always_inline)) {
    ++loaded;
    if (++ij * DWG_BK >= inb) {
      ij = 0;
      ++ib;
      while (ib < nB && (inb = __builtin_amdgcn_readlane(nb_lane, ib)) == 0) ++ib;
    }
  };
  auto store_piece = [&](unsigned char* stage, int q, u32x4 v) __attribute__((always_inline)) {
    const int p = 2 * wave + (q & 1);
    *(u32x4*)(stage + (q < 2 ? 0 : DWG_OPND) + f_lds + p * 64) = v;
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment reads (dwconv.hip): lane supplies chunk (l15 & 3) of row r0 + 8*h of the tile's image
  const int r0 = 16 * (lg >> 1) + 4 * (lg & 1) + (l15 >> 2), cq = (l15 & 3) << 3;
  const int a_lane = (wm * 4) * DWG_IMG + r0 * 32 + cq;
  const int b_lane = DWG_OPND + (wn * 8) * DWG_IMG + r0 * 32 + cq;
  typedef __attribute__((address_space(3))) bf16x4* tr_ptr;
  auto read_a = [&](const unsigned char* st, int i) __attribute__((always_inline)) -> bf16x8 {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + a_lane + i * DWG_IMG));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + a_lane + i * DWG_IMG + 256));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto read_b = [&](const unsigned char* st, int j) __attribute__((always_inline)) -> bf16x8 {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + b_lane + j * DWG_IMG));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(st + b_lane + j * DWG_IMG + 256));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

#define DWG_SB() __builtin_amdgcn_sched_barrier(0)
  // The barrier is issued from asm: for `__builtin_amdgcn_s_barrier` hipcc drains every counter first, the loads in flight included.
#define DWG_BARRIER() asm volatile("s_barrier" ::: "memory")
  if (nsteps > 0) {
    // set[n & 1] carries K step n from its loads (issued during step n - 4) to its LDS stores (during step n - 2) into stage n % 3; step
    // n reads that stage after the barrier that ends step n - 1.  Step s therefore stores step s + 2 from set[s & 1] and refills the set
    // with the loads of step s + 4.  The stage written during step s, (s + 2) % 3, was last read during step s - 1 (and by the fragment
    // prefetch of step s - 2): all before the barrier that ended step s - 1.
    u32x4 set[2][4];
    const int nsteps2 = (nsteps + 1) & ~1;
    // prologue: steps 0..3 loaded (0, 2 -> set 0 in turn; 1, 3 -> set 1), steps 0 and 1 stored
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) set[n][q] = load_piece(q, loaded < nsteps);
      if (loaded < nsteps) load_advance();
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) store_piece(smem + n * DWG_STAGE, q, set[n][q]);
      const bool more = loaded < nsteps;
#pragma unroll
      for (int q = 0; q < 4; ++q) set[n][q] = load_piece(q, more);
      if (more) load_advance();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DWG_BARRIER();
    const unsigned char* st = smem;                  // stage of the current step
    unsigned char* stw = smem + 2 * DWG_STAGE;       // stage written during the current step: (s + 2) % 3
    bf16x8 af[2][4], bf[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = read_a(st, i);
    bf[0] = read_b(st, 0);
    bf[1] = read_b(st, 1);
    auto do_step = [&](int par, bf16x8 (&ac)[4], bf16x8 (&an)[4]) __attribute__((always_inline)) {
      const bool more = loaded < nsteps;
      const unsigned char* stn = (st == smem + 2 * DWG_STAGE) ? smem : st + DWG_STAGE;
      DWG_SB();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 2 < 8) bf[(j + 2) % 3] = read_b(st, j + 2);
        if ((j & 1) == 0) {
          const int q = j >> 1;
          store_piece(stw, q, set[par][q]);
          set[par][q] = load_piece(q, more);
          if (q == 3 && more) load_advance();
        }
        // the next step's first fragments: its stage was completed before the barrier that ended the previous step
        if (j >= 2 && j < 6) an[j - 2] = read_a(stn, j - 2);
        if (j >= 6) bf[(j + 2) % 3] = read_b(stn, j + 2 - 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = mfma16<false>(ac[i], bf[j % 3], acc[i][j]);
        DWG_SB();
      }
      DWG_BARRIER();
      st = stn;
      stw = (stw == smem + 2 * DWG_STAGE) ? smem : stw + DWG_STAGE;
    };
    // 8 % 3 = 2: the ring position of bf advances by 2 per step; two steps per iteration keep it a compile-time pattern only if the body
    // re-bases it — so each copy of the body starts by rotating bf back to (0, 1)
#pragma unroll 1
    for (int step = 0; step < nsteps2; step += 2) {
      do_step(0, af[0], af[1]);
      { const bf16x8 t0 = bf[2], t1 = bf[0]; bf[0] = t0; bf[1] = t1; }
      do_step(1, af[1], af[0]);
      { const bf16x8 t0 = bf[2], t1 = bf[0]; bf[0] = t0; bf[1] = t1; }
    }
  }
#undef DWG_SB
#undef DWG_BARRIER

  // ---- epilogue: the wave's 64 x 128 tile (lane: rows 4*lg .. +3, column l15 of each 16 x 16 fragment)
  const int Cin = P.Cin, Cout = P.Cout, taps = P.taps;
  const bool slab = P.splits > 1;
  float* base = slab ? P.out + ((int64_t)(split * taps + tap) * Cout) * Cin : P.out + (int64_t)tap * Cin;
  const int64_t rstride = slab ? (int64_t)Cin : (int64_t)taps * Cin;
  const bool accumulate = !slab && P.accumulate;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* row = base + (int64_t)(m0 + wm * 64 + i * 16 + 4 * lg + r) * rstride + n0 + wn * 128 + l15;
      if (accumulate) {
        float old[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) old[j] = row[j * 16];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j * 16] = old[j] + acc[i][j][r];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j * 16] = acc[i][j][r];
      }
    }
  }
}

// The grid may be smaller than the number of tiles (max_wgs: one workgroup per CU on part of the chip, the rest left to a concurrent
// stream): a workgroup then walks tiles w, w + gridDim.x, ... in XCD-major order (the tiles that share a dY tile and K range meet in
// one L2 when the grid is a multiple of 8).
__global__ __launch_bounds__(DWG_NT, 1) void dwgemm_kernel(const DwgArgs args, int total) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DWG_SMEM];
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int q = total >> 3, r = total & 7, x = w & 7;
    const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
    dwgemm_tile(args, id, smem);
    __syncthreads();                   // the next tile's prologue overwrites stages this tile's slower waves may still read
  }
}

}  // namespace

extern "C" int ttsk_dwgemm_supported(int Cout, int Cin, int K) {
  return K >= 1 && K <= 15 && (K & 1) == 1 && Cout > 0 && Cin > 0 && Cout % 256 == 0 && Cin % 256 == 0;
}

extern "C" int64_t ttsk_dwgemm_workspace_floats(int Cout, int Cin, int K, int splits) {
  return splits > 1 ? (int64_t)splits * K * Cout * Cin : 0;
}

extern "C" int ttsk_dwgemm_batch(const ttsk_dwgemm_item* items, int n, int max_wgs, void* stream) {
  TTSK_REQUIRE(items && n > 0 && n <= DWG_MAXP, "dwgemm_batch: 1..%d items", DWG_MAXP);
  DwgArgs a;
  a.n = n;
  int wg = 0;
  for (int i = 0; i < n; ++i) {
    const ttsk_dwgemm_item& it = items[i];
    TTSK_REQUIRE(it.dy && it.x && it.dw, "dwgemm_batch: null pointer in item %d", i);
    TTSK_REQUIRE(ttsk_dwgemm_supported(it.Cout, it.Cin, it.K), "dwgemm_batch: no instance for Cout=%d Cin=%d K=%d", it.Cout, it.Cin, it.K);
    TTSK_REQUIRE(it.B > 0 && it.S > 0 && it.splits >= 1 && it.splits <= it.B && (it.B + it.splits - 1) / it.splits <= 64,
                 "dwgemm_batch: B=%d utterances in %d splits (at most 64 per split)", it.B, it.splits);
    TTSK_REQUIRE(it.splits == 1 || it.workspace, "dwgemm_batch: splits > 1 needs a workspace of ttsk_dwgemm_workspace_floats");
    TTSK_REQUIRE(it.ldy >= it.Cout && it.ldx >= it.Cin && (it.ldy & 7) == 0 && (it.ldx & 7) == 0, "dwgemm_batch: row pitches (multiples of 8, >= channels)");
    TTSK_REQUIRE((int64_t)it.B * it.S * (it.ldy > it.ldx ? it.ldy : it.ldx) * 2 < 0x7FFFFFF0ll, "dwgemm_batch: operand beyond the 2 GiB buffer range");
    TTSK_REQUIRE(((((uintptr_t)it.dy) | ((uintptr_t)it.x) | ((uintptr_t)it.dw) | ((uintptr_t)it.workspace)) & 15) == 0, "dwgemm_batch: 16-byte alignment");
    DwgProblem& p = a.p[i];
    p.dy = (const bf16_t*)it.dy; p.x = (const bf16_t*)it.x; p.lens = (const long long*)it.lens;
    p.out = it.splits > 1 ? it.workspace : it.dw;
    p.Cout = it.Cout; p.Cin = it.Cin; p.taps = it.K; p.ldy = it.ldy; p.ldx = it.ldx; p.B = it.B; p.S = it.S;
    p.accumulate = it.accumulate; p.splits = it.splits;
    p.tiles_m = it.Cout / 256; p.tiles_n = it.Cin / 256;
    p.nwg = p.tiles_m * p.tiles_n * it.K * it.splits;
    p.wg0 = wg;
    wg += p.nwg;
  }
  TTSK_REQUIRE(max_wgs >= 0, "dwgemm_batch: max_wgs >= 0 (0 = a workgroup per tile)");
  hipLaunchKernelGGL(dwgemm_kernel, dim3(max_wgs > 0 && max_wgs < wg ? max_wgs : wg), dim3(DWG_NT), 0, (hipStream_t)stream, a, wg);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
