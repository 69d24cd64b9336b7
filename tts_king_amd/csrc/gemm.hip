// gemm.hip — the one MFMA contraction kernel behind every GEMM-shaped op on the path (see include/ttsk.h).
//
// Tile 128(M) x 128(N) x 64(K) per 256-thread workgroup = 4 waves (2x2), each wave a 64x64 sub-tile made of
// 4x4 v_mfma_f32_16x16x32_bf16 accumulators (64 accumulator VGPRs).  Operand tiles go global -> registers ->
// LDS (16 B per lane, XOR-swizzled so that both the ds_write_b128 and the fragment reads are conflict-free),
// double-buffered with the next tile's global loads in flight during the MFMAs, one barrier per K tile.
// Operands whose contraction index is the memory ROW (dW, Pᵀ·dO, dSᵀ·Q ...) are staged untransposed and read
// with ds_read_b64_tr_b16, the gfx950 transposing LDS read, so no transposed copy of any tensor ever exists.
// Conv1d is an implicit GEMM: the K loop walks (tap, channel-chunk) and a tap only shifts the A row index
// (channels-last activations), with zero fill outside the utterance.  The epilogue goes through LDS so that
// C (and the residual / gate operands) move as full 16-byte rows.
#include <cstdlib>
#include <cstring>
#include "gemm_common.h"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int CS_LD = 132;                               // fp32 epilogue tile leading dim (floats)
// BM = 128: the default tile.  BM = 64 (operand A untransposed only): the same kernel with half the rows per workgroup,
// for outputs with few columns (N = 256: 6768 rows give 106 tiles of 128^2 for 256 CUs, which is what used to force
// split-K plus a reducer launch on the forward / dX chain; 64-row tiles give 212 workgroups and need neither).
template <int BM> struct TileCfg {
  static constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;                           // 32 / 24 KiB
  static constexpr int EPI_BYTES = BM * CS_LD * 4;                                      // 67,584 / 33,792 B
  static constexpr int SMEM_BYTES = EPI_BYTES > 2 * STAGE_BYTES ? EPI_BYTES : 2 * STAGE_BYTES;
  static constexpr int AI = BM / 32;      // 16-byte A chunks per thread per K tile
  static constexpr int WM = BM / 2;       // rows of a wave's sub-tile (waves 2 x 2)
  static constexpr int AM = WM / 16;      // MFMA row tiles per wave
};

typedef GemmArgs Args;


// one output tile (bid_in of the problem's tiles_m*tiles_n, batch index z of nzgrid, K range `split`)
template <int BM, bool ATR, bool BTR, bool F16>
__device__ __forceinline__ void gemm_tile(const Args& g, int bid_in, int z, int split, int nzgrid, unsigned char* smem) {
  static_assert(BM == 128 || !ATR, "the 64-row tile is instantiated for untransposed A only");
  constexpr int STAGE_BYTES = TileCfg<BM>::STAGE_BYTES, AI = TileCfg<BM>::AI, WM = TileCfg<BM>::WM, AM = TileCfg<BM>::AM;
  const ttsk_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile id with XCD-aware (bijective) remap: consecutive logical tiles share an XCD's L2
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = bid_in;
  {
    const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z1 = z / d.nz2, z2 = z - z1 * d.nz2;

  const bf16_t* __restrict__ A = (const bf16_t*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const bf16_t* __restrict__ B = (const bf16_t*)d.B + z1 * d.sB1 + z2 * d.sB2;
  // Buffer descriptors: operands are read with buffer_load_dwordx4 and a 32-bit byte offset; every predicate
  // (M/N/K tails, conv zero padding outside the utterance) becomes an out-of-range offset, for which the hardware
  // returns zeros — no branches and no selects in the K loop.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7FFFFFF0, 0x00020000);
  constexpr int OOB = 0x7FFFFFFF;
  const int M = d.M, N = d.N, K = d.K;
  const int taps = d.taps > 0 ? d.taps : 1;
  const bool conv_a = d.taps > 0;
  const int bshift = d.bseg_len > 0 ? d.bshift0 + z2 * d.bdshift : 0;
  const int K8 = (K + 7) & ~7;

  // ---- per-thread staging coordinates (4 x 16 B per operand per K tile)
  // normal operand tile [128 rows][64 k]: chunk c = tid + 256 i -> row = c >> 3, slot = c & 7
  // transposed operand tile [64 k][128 m]: chunk c -> krow = c >> 4, slot16 = c & 15
  // Everything that does not change along K is folded into per-chunk values here (byte offset at k = 0 / tap 0, a
  // validity bit per tap for the conv zero padding), so the K loop spends a few VALU instructions per load.
  const int nrow = tid >> 3, nslot = tid & 7;      // + 32 i rows
  const int trow = tid >> 4, tslot = tid & 15;     // + 16 i krows
  int a_off[AI], b_off[4], b_tt[4];
  unsigned a_ok[AI], b_ok[4];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    if (!ATR) {
      const int gm = m0 + nrow + 32 * i;
      a_off[i] = (gm * d.lda + nslot * 8) * 2;
      unsigned okm = 0;
      if (gm < M) {
        if (conv_a) {
          const int t = gm % d.seg_len;
          for (int tp = 0; tp < taps; ++tp) {
            const int tt = t + d.tap_shift0 + tp * d.tap_dshift;
            okm |= (tt >= 0 && tt < d.seg_len) ? (1u << (tp & 31)) : 0u;
          }
        } else okm = 1u;
      }
      a_ok[i] = okm;
    } else {
      const int mcol = m0 + tslot * 8;
      a_off[i] = ((trow + 16 * i) * d.lda + mcol) * 2;
      a_ok[i] = mcol < M ? 1u : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!BTR) {
      const int gn = n0 + nrow + 32 * i;
      b_off[i] = (gn * d.ldb + nslot * 8) * 2;
      b_ok[i] = gn < N ? 1u : 0u;
    } else {
      const int ncol = n0 + tslot * 8;
      b_off[i] = ((trow + 16 * i) * d.ldb + ncol) * 2;
      b_ok[i] = ncol < N ? 1u : 0u;
    }
    b_tt[i] = 0;
  }

  const int kc_begin = split * g.chunks_per_split;
  int kc_end = kc_begin + g.chunks_per_split;
  if (kc_end > g.kchunks) kc_end = g.kchunks;
  const int per = kc_end > kc_begin ? kc_end - kc_begin : 0;
  const int nk = per * taps;

  const float in_slope = d.in_slope;
  const bool lrelu_in = d.flags & TTSK_GEMM_LRELU_IN;
  const bool bseg = BTR && d.bseg_len > 0;
  const bool wide_taps = taps > 32;                 // the per-tap validity bits cover 32 taps; beyond that, test on the fly

  // load state: the next tile to load is chunk kc_begin + ld_kk of tap ld_tap
  int ld_tap = 0, ld_kk = 0;
  auto reset_btt = [&]() __attribute__((always_inline)) {   // position inside its utterance of the chunk's k-row (dW of a conv)
    if (bseg) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b_tt[i] = (kc_begin * BK + trow + 16 * i) % d.bseg_len;
    }
  };
  reset_btt();

  struct Regs { uint4 a[AI], b[4]; };
#define TTSK_LD(rs, off) __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0))
  // `live` = the tile exists (a step past the last tile still issues its loads, all out of range: zeros, no branch)
  auto load_tile = [&](Regs& R, bool live) __attribute__((always_inline)) {
    const int kbase = (kc_begin + ld_kk) * BK;
    int oa[AI], ob[4];
    if (!ATR) {
      const int shift = conv_a ? d.tap_shift0 + ld_tap * d.tap_dshift : 0;
      const int add = (shift * d.lda + kbase) * 2;
      const bool kok = live && kbase + nslot * 8 < K8;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        bool ok = kok && ((a_ok[i] >> (ld_tap & 31)) & 1u);
        if (wide_taps) {   // recompute exactly (rare: > 32 taps)
          const int tt = (m0 + nrow + 32 * i) % d.seg_len + shift;
          ok = kok && a_ok[i] != 0 && tt >= 0 && tt < d.seg_len;
        }
        oa[i] = ok ? a_off[i] + add : OOB;
      }
    } else {
      const int add = kbase * d.lda * 2;
#pragma unroll
      for (int i = 0; i < AI; ++i) oa[i] = (live && kbase + trow + 16 * i < K && a_ok[i]) ? a_off[i] + add : OOB;
    }
    const int tapoff = ld_tap * (int)d.b_tap_stride;
    if (!BTR) {
      const bool kok = live && kbase + nslot * 8 < K8;
      const int add = (tapoff + kbase) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) ob[i] = (kok && b_ok[i]) ? b_off[i] + add : OOB;
    } else {
      const int add = ((kbase + bshift) * d.ldb + tapoff) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bool ok = live && kbase + trow + 16 * i < K && b_ok[i];
        if (bseg) ok = ok && (unsigned)(b_tt[i] + bshift) < (unsigned)d.bseg_len;
        ob[i] = ok ? b_off[i] + add : OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) R.a[i] = TTSK_LD(rsA, oa[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) R.b[i] = TTSK_LD(rsB, ob[i]);
    // advance to the next tile
    if (++ld_kk == per) {
      ld_kk = 0;
      ++ld_tap;
      reset_btt();
    } else if (bseg) {
      if (d.bseg_len >= BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { b_tt[i] += BK; b_tt[i] -= b_tt[i] >= d.bseg_len ? d.bseg_len : 0; }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) b_tt[i] = (b_tt[i] + BK) % d.bseg_len;
      }
    }
  };
#undef TTSK_LD

  auto store_tile = [&](int buf, Regs& R) __attribute__((always_inline)) {
    unsigned char* sa = smem + buf * STAGE_BYTES;
    unsigned char* sb = sa + BM * BK * 2;
    if (lrelu_in && !ATR) {
#pragma unroll
      for (int i = 0; i < AI; ++i) R.a[i] = lrelu8<F16>(R.a[i], in_slope);
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      if (!ATR) { const int row = nrow + 32 * i; *(uint4*)(sa + row * 128 + ((nslot ^ (row & 7)) << 4)) = R.a[i]; }
      else { const int kr = trow + 16 * i; *(uint4*)(sa + kr * 256 + (((tslot >> 1) ^ tr_sw(kr)) << 5) + ((tslot & 1) << 4)) = R.a[i]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!BTR) { const int row = nrow + 32 * i; *(uint4*)(sb + row * 128 + ((nslot ^ (row & 7)) << 4)) = R.b[i]; }
      else { const int kr = trow + 16 * i; *(uint4*)(sb + kr * 256 + (((tslot >> 1) ^ tr_sw(kr)) << 5) + ((tslot & 1) << 4)) = R.b[i]; }
    }
  };

  f32x4 acc[AM][4];
#pragma unroll
  for (int i = 0; i < AM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lg = lane >> 4;

  auto compute_tile = [&](int buf) {
    const unsigned char* sa = smem + buf * STAGE_BYTES;
    const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[AM], bfr[4];
#pragma unroll
      for (int i = 0; i < AM; ++i) {
        if (!ATR) {
          const int row = wm * WM + i * 16 + l15;
          af[i] = *(const bf16x8*)(sa + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        } else {
          // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block and receives
          // column (lane & 15) of its 4 rows: element j <- tile[k = 8*lg + j][m = block col]
          const int mblk = (wm * WM + i * 16) >> 4;  // 32-byte slot index of the 16-column block
          const int k0 = ks * 32 + 8 * lg + (l15 >> 2);
          const int k1 = k0 + 4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sa + k0 * 256 + ((mblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sa + k1 * 256 + ((mblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
          af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!BTR) {
          const int row = wn * 64 + i * 16 + l15;
          bfr[i] = *(const bf16x8*)(sb + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        } else {
          const int nblk = (wn * 64 + i * 16) >> 4;
          const int k0 = ks * 32 + 8 * lg + (l15 >> 2);
          const int k1 = k0 + 4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sb + k0 * 256 + ((nblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sb + k1 * 256 + ((nblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
          bfr[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16<F16>(af[i], bfr[j], acc[i][j]);
    }
  };

  // ---- K loop: two register sets, so the global loads of tiles kt+1 and kt+2 are in flight while tile kt is
  // multiplied (a lone workgroup on its CU was latency-bound at one tile in flight: 1.0 us per K step against 0.25 us
  // of MFMA); LDS is double-buffered, one barrier per K tile.
  // One step = global loads of tile kt+2, MFMAs of tile kt, LDS writes of tile kt+1 — issued by ONE wave per SIMD, so they
  // only overlap if the instruction stream interleaves them: the sched_group_barrier sequence asks for "LDS reads of a
  // contraction half, then 2 MFMAs : 1 global load (first half) / 2 MFMAs : 1 LDS write (second half)".  Loads past the
  // last tile are out-of-range buffer reads (zeros) and the write past it goes to the buffer nobody reads again: the
  // step is branch-free, one scheduling region.
  if (nk > 0) {
    Regs R0, R1;
    load_tile(R0, true);
    load_tile(R1, nk > 1);
    store_tile(0, R0);
    __syncthreads();
    constexpr int NRD = (ATR ? 2 * AM : AM) + (BTR ? 8 : 4);     // LDS read instructions per contraction half
    constexpr int NLD = AI + 4;                                  // global loads = LDS writes per K tile
    constexpr int MPG = (4 * AM) / NLD > 0 ? (4 * AM) / NLD : 1; // MFMAs between two loads / two LDS writes
#define TTSK_STEP(CUR, RL, RS)                                                      \
    {                                                                               \
      load_tile(RL, kt + 2 + CUR < nk);                                             \
      compute_tile(CUR);                                                            \
      store_tile(1 - CUR, RS);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);                          \
      _Pragma("unroll") for (int q = 0; q < NLD; ++q) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);                        \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                          \
      }                                                                             \
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);                          \
      _Pragma("unroll") for (int q = 0; q < NLD; ++q) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);                        \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                          \
      }                                                                             \
      __syncthreads();                                                              \
    }
    for (int kt = 0; kt < nk; kt += 2) {
      TTSK_STEP(0, R0, R1)
      if (kt + 1 < nk) TTSK_STEP(1, R1, R0)
    }
#undef TTSK_STEP
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> full-row 16-byte traffic
  float* cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < AM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cs[(wm * WM + i * 16 + lg * 4 + r) * CS_LD + wn * 64 + j * 16 + l15] = acc[i][j][r];
  __syncthreads();

  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;
  const int64_t roff = z1 * d.sR1 + z2 * d.sR2;
  const int cg = tid & 15;
  const int gn = n0 + cg * 8;
  if (gn >= N) return;
  const int nvalid = (N - gn) < 8 ? (N - gn) : 8;
  if (d.splits > 1 || (d.flags & TTSK_GEMM_RAW_SLABS)) {
    // split-K: raw fp32 partial sums into the workspace slab [split][z][M][N]; the reducer applies the epilogue
    float* ws = (float*)d.workspace + ((int64_t)split * nzgrid + z) * ((int64_t)M * N);
    for (int p = 0; p < BM / 16; ++p) {
      const int row = p * 16 + (tid >> 4);
      const int gm = m0 + row;
      if (gm >= M) continue;
      float* wp = ws + (int64_t)gm * N + gn;
      const f32x4 lo = *(const f32x4*)(cs + row * CS_LD + cg * 8);
      const f32x4 hi = *(const f32x4*)(cs + row * CS_LD + cg * 8 + 4);
      if (nvalid == 8 && (N & 3) == 0) {
        *(f32x4*)wp = lo;
        *(f32x4*)(wp + 4) = hi;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { if (e < nvalid) wp[e] = lo[e]; if (e + 4 < nvalid) wp[e + 4] = hi[e]; }
      }
    }
    return;
  }
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (d.bias && e < nvalid) ? d.bias[z1 * d.s_bias1 + gn + e] : 0.f;
  // residual / gate operands of all this thread's row segments first (independent loads in flight together), then the stores
  constexpr int NP = BM / 16;
  EpiOperands ops[NP];
  int64_t orow[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int gm = m0 + p * 16 + (tid >> 4);
    orow[p] = gm < M ? epilogue_row(d, gm, z2) : -1;
    ops[p] = epilogue_load(d, roff, orow[p], gn, nvalid);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int row = p * 16 + (tid >> 4);
    if (orow[p] < 0) continue;
    float v[8];
    {
      const f32x4 lo = *(const f32x4*)(cs + row * CS_LD + cg * 8);
      const f32x4 hi = *(const f32x4*)(cs + row * CS_LD + cg * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[e + 4] = hi[e]; }
    }
    epilogue_apply<F16>(d, coff, roff, orow[p], gn, nvalid, v, bias, ops[p]);
  }
}

template <bool ATR, bool BTR, bool F16>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TileCfg<128>::SMEM_BYTES];
  gemm_tile<128, ATR, BTR, F16>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, smem);
}
// the 64-row tile (desc.kernel = 3)
template <bool BTR, bool F16>
__global__ __launch_bounds__(NTHREADS, 2) void gemm64_kernel(const Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TileCfg<64>::SMEM_BYTES];
  gemm_tile<64, false, BTR, F16>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, smem);
}

// Grouped launch: the workgroups of n independent problems (same operand layout) as ONE grid.  A training step's weight-
// gradient GEMMs have 32-256 workgroups each — a fraction of the 512 resident slots — and nothing but Adam waits for
// them, so they are queued during backward and share one launch: 42 launches of 6-22 us (605 us) become one.
// prefix[p] = first workgroup of problem p; a workgroup finds its problem by bisection and reads the problem's arguments
// from the table with scalar loads (the table pointer and p are wave-uniform).
template <bool ATR, bool BTR, bool F16>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_group_kernel(const int* __restrict__ prefix, const Args* __restrict__ args, int n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TileCfg<128>::SMEM_BYTES];
  const int wg = blockIdx.x;
  int lo = 0, hi = n;                 // invariant: prefix[lo] <= wg < prefix[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= wg) lo = mid; else hi = mid;
  }
  const int p = __builtin_amdgcn_readfirstlane(lo);
  const Args& g = args[p];
  const int local = wg - prefix[p];
  const int tiles = g.tiles_m * g.tiles_n, nz = g.d.nz1 * g.d.nz2;
  const int tile = local % tiles, rest = local / tiles;
  gemm_tile<128, ATR, BTR, F16>(g, tile, rest % nz, rest / nz, nz, smem);
}

// split-K reducer: sums the `splits` workspace slabs in fixed order and applies the epilogue (deterministic)
template <bool F16>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const Args g) {
  const ttsk_gemm_desc& d = g.d;
  const int M = d.M, N = d.N;
  const int ngrp = (N + 7) >> 3;
  const int z = blockIdx.y;
  const int z1 = z / d.nz2, z2 = z - z1 * d.nz2;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;
  const int64_t roff = z1 * d.sR1 + z2 * d.sR2;
  const int64_t mn = (int64_t)M * N;
  const float* ws = (const float*)d.workspace + (int64_t)z * mn;
  const int64_t sstride = (int64_t)gridDim.y * mn;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)M * ngrp; i += (int64_t)gridDim.x * 256) {
    const int gm = (int)(i / ngrp);
    const int gn = (int)(i - (int64_t)gm * ngrp) * 8;
    const int nvalid = (N - gn) < 8 ? (N - gn) : 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p = ws + (int64_t)gm * N + gn;
    if (nvalid == 8 && (N & 3) == 0) {
      for (int s = 0; s < d.splits; ++s) {
        const f32x4 lo = *(const f32x4*)(p + s * sstride), hi = *(const f32x4*)(p + s * sstride + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[e + 4] += hi[e]; }
      }
    } else {
      for (int s = 0; s < d.splits; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += p[s * sstride + e];
    }
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = (d.bias && e < nvalid) ? d.bias[z1 * d.s_bias1 + gn + e] : 0.f;
    epilogue_store<F16>(d, coff, roff, gm, gn, nvalid, v, bias, z2);
  }
}

}  // namespace

// ---- deferred split-K reduction of weight-gradient GEMMs: one launch sums the slabs of up to 64 GEMMs
struct ReduceBatch {
  ttsk_reduce_item it[64];
  int n;
};

namespace {
__global__ __launch_bounds__(256) void gemm_reduce_batch_kernel(const ReduceBatch rb) {
  const ttsk_reduce_item& it = rb.it[blockIdx.y];
  const int M = it.M, N = it.N;
  const int ngrp = (N + 7) >> 3;
  const int64_t mn = (int64_t)M * N;
  const int64_t sstride = (int64_t)it.nz * mn;
  const bool vec = (N & 3) == 0 && (it.ldc & 3) == 0 && (it.sC2 & 3) == 0;
  const int64_t total = (int64_t)it.nz * M * ngrp;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int z = (int)(i / ((int64_t)M * ngrp));
    const int64_t r = i - (int64_t)z * M * ngrp;
    const int gm = (int)(r / ngrp);
    const int gn = (int)(r - (int64_t)gm * ngrp) * 8;
    const int nvalid = (N - gn) < 8 ? (N - gn) : 8;
    const float* p = it.ws + (int64_t)z * mn + (int64_t)gm * N + gn;
    float* cp = it.C + (int64_t)z * it.sC2 + (int64_t)gm * it.ldc + gn;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (nvalid == 8 && vec) {
      for (int s = 0; s < it.splits; ++s) {
        const f32x4 lo = *(const f32x4*)(p + s * sstride), hi = *(const f32x4*)(p + s * sstride + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[e + 4] += hi[e]; }
      }
      f32x4 lo = f32x4{v[0], v[1], v[2], v[3]} * it.alpha, hi = f32x4{v[4], v[5], v[6], v[7]} * it.alpha;
      if (it.accumulate) { lo += *(const f32x4*)cp; hi += *(const f32x4*)(cp + 4); }
      *(f32x4*)cp = lo;
      *(f32x4*)(cp + 4) = hi;
    } else {
      for (int s = 0; s < it.splits; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += p[s * sstride + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = (it.accumulate ? cp[e] : 0.f) + v[e] * it.alpha;
    }
  }
}
}  // namespace

extern "C" int ttsk_gemm_reduce_batch(const ttsk_reduce_item* items, int n, void* stream) {
  TTSK_REQUIRE(items && n > 0, "ttsk_gemm_reduce_batch: bad arguments");
  for (int base = 0; base < n; base += 64) {
    ReduceBatch rb;
    rb.n = n - base < 64 ? n - base : 64;
    for (int i = 0; i < rb.n; ++i) {
      rb.it[i] = items[base + i];
      TTSK_REQUIRE(rb.it[i].ws && rb.it[i].C && rb.it[i].M > 0 && rb.it[i].N > 0 && rb.it[i].splits > 0 && rb.it[i].nz > 0,
                   "ttsk_gemm_reduce_batch: bad item %d", base + i);
    }
    hipLaunchKernelGGL(gemm_reduce_batch_kernel, dim3(96, rb.n), dim3(256), 0, (hipStream_t)stream, rb);
    TTSK_CHECK_LAUNCH();
  }
  return TTSK_OK;
}

// ---- planning: which tile configuration, how many K splits
namespace {
struct Plan { int kernel, splits, tiles_m, tiles_n, kchunks, chunks_per_split; int64_t ws_bytes; };

int validate(ttsk_gemm_desc& d) {
  TTSK_REQUIRE(d.A && d.B && d.C, "ttsk_gemm: null operand");
  TTSK_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "ttsk_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  const bool atr = d.flags & TTSK_GEMM_A_TR, btr = d.flags & TTSK_GEMM_B_TR, f16 = d.flags & TTSK_GEMM_F16;
  TTSK_REQUIRE(!(atr && !btr), "ttsk_gemm: A_TR without B_TR is not instantiated");
  TTSK_REQUIRE(!(f16 && atr), "ttsk_gemm: fp16 operands are instantiated for the inference shapes only (A untransposed)");
  TTSK_REQUIRE((d.lda & 7) == 0 && (d.ldb & 7) == 0, "ttsk_gemm: lda/ldb must be multiples of 8 (16-byte rows)");
  TTSK_REQUIRE((((uintptr_t)d.A) & 15) == 0 && (((uintptr_t)d.B) & 15) == 0, "ttsk_gemm: A/B must be 16-byte aligned");
  TTSK_REQUIRE(((d.sA1 | d.sA2 | d.sB1 | d.sB2) & 7) == 0, "ttsk_gemm: batch strides of A/B must be multiples of 8");
  TTSK_REQUIRE(d.taps == 0 || (!atr && d.seg_len > 0 && d.M % d.seg_len == 0), "ttsk_gemm: conv-A mode needs A untransposed and M %% seg_len == 0");
  TTSK_REQUIRE(d.taps == 0 || (d.b_tap_stride & 7) == 0, "ttsk_gemm: b_tap_stride must be a multiple of 8");
  TTSK_REQUIRE(d.bseg_len == 0 || btr, "ttsk_gemm: B row shift needs B_TR");
  TTSK_REQUIRE(d.out_mul == 0 || (d.seg_len > 0 && d.out_seg > 0), "ttsk_gemm: output remap needs seg_len/out_seg");
  TTSK_REQUIRE(d.kernel >= 0 && d.kernel <= 3 && d.splits >= 0, "ttsk_gemm: kernel must be 0 (auto), 1, 2 or 3; splits >= 0");
  TTSK_REQUIRE(!(d.kernel == 3 && atr), "ttsk_gemm: kernel = 3 (64-row tile) is instantiated for untransposed A only");
  TTSK_REQUIRE(!(d.kernel == 2 && (d.flags & TTSK_GEMM_LRELU_IN)), "ttsk_gemm: LRELU_IN needs the register-staged kernel (kernel = 1)");
  TTSK_REQUIRE(!(d.kernel == 2 && d.taps > 32), "ttsk_gemm: kernel = 2 handles at most 32 taps");
  if (d.nz1 < 1) d.nz1 = 1;
  if (d.nz2 < 1) d.nz2 = 1;
  {
    // operands are addressed with 32-bit BYTE offsets from the per-batch origin (buffer_load ... offen): the extent a launch
    // can touch, conv tap shifts and tap-major weight strides included, must stay below 2^31
    const int64_t taps = d.taps > 0 ? d.taps : 1;
    const int64_t a_rows = atr ? d.K : d.M, a_cols = atr ? d.M : d.K;
    const int64_t b_rows = btr ? d.K : d.N, b_cols = btr ? d.N : d.K;
    const int64_t a_shift = d.taps > 0 ? (int64_t)llabs((long long)d.tap_shift0) + taps * llabs((long long)d.tap_dshift) : 0;
    const int64_t b_shift = d.bseg_len > 0 ? (int64_t)llabs((long long)d.bshift0) + (int64_t)d.nz2 * llabs((long long)d.bdshift) : 0;
    const int64_t a_ext = ((a_rows + a_shift + 256) * d.lda + a_cols + 64) * 2;
    const int64_t b_ext = ((b_rows + b_shift + 256) * d.ldb + b_cols + 64 + taps * llabs((long long)d.b_tap_stride)) * 2;
    TTSK_REQUIRE(a_ext < ((int64_t)1 << 31) && b_ext < ((int64_t)1 << 31),
                 "ttsk_gemm: operand extent exceeds the 2 GiB a 32-bit buffer offset reaches (A %lld B, B %lld B): split the batch",
                 (long long)a_ext, (long long)b_ext);
  }
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_ADD_R) || d.R, "ttsk_gemm: ADD_R without R");
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_MASK_G) || d.G, "ttsk_gemm: MASK_G without G");
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_ACCUM_C) || (d.flags & TTSK_GEMM_C_F32), "ttsk_gemm: ACCUM_C needs fp32 C");
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_RAW_SLABS) || (d.out_mul == 0 && !d.C2), "ttsk_gemm: RAW_SLABS takes no output remap / second output");
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_DEFER_REDUCE) ||
                   ((d.flags & TTSK_GEMM_C_F32) && !d.bias && !d.R && !d.G && !d.C2 && d.out_mul == 0 && d.nz1 <= 1 &&
                    !(d.flags & (TTSK_GEMM_RELU | TTSK_GEMM_TANH | TTSK_GEMM_LRELU_OUT))),
               "ttsk_gemm: DEFER_REDUCE is for plain fp32 outputs (weight gradients) only");
  return TTSK_OK;
}

// Cost model in microseconds, calibrated on MI355X with tools/debug/gemm_tune.py: a K step of a 128^2 workgroup takes
// 1.1 us alone on its CU and 1.5 us when two share it, a K step of the 256x128 ring 1.3 us; SLOTS workgroups run at once;
// a split-K reducer costs a launch plus its slab traffic.  Small problems (the encoder's 1024 rows) gain most from
// splitting K (16 tiles -> 256 workgroups: 164 us -> 26 us for the k=9 conv dX), large ones take the big tile.
constexpr float T_ITER1 = 1.5f, T_ITER1_ALONE = 1.1f, T_ITER2 = 1.3f, T_ITER3 = 1.0f, T_ITER3_ALONE = 0.7f, T_ITER3_CROWDED = 1.35f, T_LAUNCH = 4.f, T_REDUCE = 7.f;
constexpr int SLOTS1 = 512, SLOTS2 = 256, SLOTS3 = 768;

Plan make_plan(const ttsk_gemm_desc& d) {
  const int nz = d.nz1 * d.nz2;
  const int taps = d.taps > 0 ? d.taps : 1;
  const int kchunks = (d.K + BK - 1) / BK;
  Plan best{};
  float best_t = -1.f;
  for (int kernel = 1; kernel <= 3; ++kernel) {
    if (d.kernel != 0 && d.kernel != kernel) continue;
    if (kernel == 2 && ((d.flags & TTSK_GEMM_LRELU_IN) || d.taps > 32)) continue;
    // the 64-row tile: for untransposed A, when the larger tiles cannot give every CU a workgroup
    if (kernel == 3 && ((d.flags & TTSK_GEMM_A_TR) || (d.kernel == 0 && (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * nz >= 256))) continue;
    const int bm = kernel == 2 ? 256 : (kernel == 3 ? 64 : 128);
    const int tm = (d.M + bm - 1) / bm, tn = (d.N + 127) / 128;
    const int64_t tiles = (int64_t)tm * tn * nz;
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    for (int ci = 0; ci < 12; ++ci) {
      int sp = cand[ci];
      if (d.splits != 0) sp = d.splits;
      if (sp > kchunks) sp = kchunks;
      const int per = (kchunks + sp - 1) / sp;
      sp = (kchunks + per - 1) / per;
      const int64_t wgs = tiles * sp;
      const int slots = kernel == 1 ? SLOTS1 : (kernel == 2 ? SLOTS2 : SLOTS3);
      const int64_t rounds = (wgs + slots - 1) / slots;
      // a K step slows down with every workgroup that shares the CU's L2 -> LDS path (64-row tile, measured in the train
      // step: 0.7 us alone, 1.3 us when three share a CU)
      const float t_iter = kernel == 2 ? T_ITER2 : kernel == 3 ? (wgs <= 256 ? T_ITER3_ALONE : wgs <= 512 ? T_ITER3 : T_ITER3_CROWDED)
                                                               : (wgs <= 256 ? T_ITER1_ALONE : T_ITER1);   // one workgroup per CU runs faster
      float t = (float)rounds * per * taps * t_iter + T_LAUNCH;
      const bool raw = d.flags & TTSK_GEMM_RAW_SLABS;
      const int64_t ws = (sp > 1 || raw) ? (int64_t)sp * nz * d.M * d.N * 4 : 0;
      if (ws > ((int64_t)512 << 20)) { if (d.splits != 0) {} else break; }
      if (sp > 1 || raw) t += (raw ? 0.f : T_REDUCE) + (float)ws * 1.25f / 4.0e6f;
      if (best_t < 0.f || t < best_t - 0.25f) {
        best_t = t;
        best = Plan{kernel, sp, tm, tn, kchunks, per, ws};
      }
      if (d.splits != 0 || sp >= kchunks) break;
    }
  }
  return best;
}
}  // namespace

extern "C" int ttsk_gemm_plan(const ttsk_gemm_desc* dp, int32_t* kernel, int32_t* splits, int64_t* workspace_bytes) {
  TTSK_REQUIRE(dp && kernel && splits && workspace_bytes, "ttsk_gemm_plan: null pointer");
  ttsk_gemm_desc d = *dp;
  const int rc = validate(d);
  if (rc != TTSK_OK) return rc;
  const Plan p = make_plan(d);
  *kernel = p.kernel;
  *splits = p.splits;
  *workspace_bytes = p.ws_bytes;
  return TTSK_OK;
}

extern "C" int ttsk_gemm(const ttsk_gemm_desc* dp, void* stream) {
  TTSK_REQUIRE(dp != nullptr, "ttsk_gemm: null descriptor");
  Args g;
  g.d = *dp;
  ttsk_gemm_desc& d = g.d;
  const int rc = validate(d);
  if (rc != TTSK_OK) return rc;
  const bool atr = d.flags & TTSK_GEMM_A_TR, btr = d.flags & TTSK_GEMM_B_TR, f16 = d.flags & TTSK_GEMM_F16;
  Plan p = make_plan(d);
  if (p.ws_bytes > 0 && !(d.workspace && d.workspace_bytes >= p.ws_bytes && (((uintptr_t)d.workspace) & 15) == 0)) {
    // auto mode without (enough) workspace: fall back to a single pass; an explicit split request is an error
    TTSK_REQUIRE(d.splits == 0 && !(d.flags & TTSK_GEMM_RAW_SLABS), "ttsk_gemm: split-K / raw slabs need a 16-byte aligned workspace of %lld bytes (got %lld)",
                 (long long)p.ws_bytes, (long long)d.workspace_bytes);
    d.splits = 1;
    p = make_plan(d);
  }
  d.kernel = p.kernel;
  d.splits = p.splits;
  g.tiles_m = p.tiles_m;
  g.tiles_n = p.tiles_n;
  g.kchunks = p.kchunks;
  g.chunks_per_split = p.chunks_per_split;
  const int nz = d.nz1 * d.nz2;
  TTSK_REQUIRE(nz <= 65535 && d.splits <= 65535, "ttsk_gemm: batch/splits too large");
  hipStream_t s = (hipStream_t)stream;
  if (d.kernel == 2) {
    ttsk_launch_gemm2(g, atr, btr, f16, s);
  } else if (d.kernel == 3) {
    dim3 grid(g.tiles_m * g.tiles_n, nz, d.splits), block(NTHREADS);
    if (btr) {
      if (f16) hipLaunchKernelGGL((gemm64_kernel<true, true>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm64_kernel<true, false>), grid, block, 0, s, g);
    } else {
      if (f16) hipLaunchKernelGGL((gemm64_kernel<false, true>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm64_kernel<false, false>), grid, block, 0, s, g);
    }
  } else {
    dim3 grid(g.tiles_m * g.tiles_n, nz, d.splits), block(NTHREADS);
    if (atr)
      hipLaunchKernelGGL((gemm_kernel<true, true, false>), grid, block, 0, s, g);
    else if (btr) {
      if (f16) hipLaunchKernelGGL((gemm_kernel<false, true, true>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_kernel<false, true, false>), grid, block, 0, s, g);
    } else {
      if (f16) hipLaunchKernelGGL((gemm_kernel<false, false, true>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_kernel<false, false, false>), grid, block, 0, s, g);
    }
  }
  TTSK_CHECK_LAUNCH();
  if (d.splits > 1 && !(d.flags & (TTSK_GEMM_DEFER_REDUCE | TTSK_GEMM_RAW_SLABS))) {
    const int64_t work = (int64_t)d.M * ((d.N + 7) / 8);
    int blocks = (int)((work + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    dim3 rgrid(blocks, nz);
    if (f16) hipLaunchKernelGGL((gemm_reduce_kernel<true>), rgrid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_reduce_kernel<false>), rgrid, dim3(256), 0, s, g);
    TTSK_CHECK_LAUNCH();
  }
  return TTSK_OK;
}

// ---------------------------------------------------------------------------------------------- grouped launch
namespace {
// Small groups (the dK / dV pair of an attention block) carry their table in the kernel arguments: no upload launch.
template <int NMAX>
struct GroupInline {
  int prefix[NMAX + 1];
  int n;
  Args args[NMAX];
};
template <bool ATR, bool BTR, bool F16, int NMAX>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_group_inline_kernel(const GroupInline<NMAX> t) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TileCfg<128>::SMEM_BYTES];
  const int wg = blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < NMAX; ++i) p += (i < t.n && t.prefix[i] <= wg) ? 1 : 0;
  const Args& g = t.args[p];
  const int local = wg - t.prefix[p];
  const int tiles = g.tiles_m * g.tiles_n, nz = g.d.nz1 * g.d.nz2;
  const int tile = local % tiles, rest = local / tiles;
  gemm_tile<128, ATR, BTR, F16>(g, tile, rest % nz, rest / nz, nz, smem);
}

// The table reaches the device through kernel arguments (3.5 KiB per launch): arguments are captured by value, so the
// grouped launch stays hipGraph-capturable without pinned host staging (pinned allocation is not permitted during capture).
struct TableChunk {
  uint4 q[224];
};
__global__ __launch_bounds__(256) void group_table_upload_kernel(const TableChunk c, uint4* __restrict__ dst, int n16) {
  if ((int)threadIdx.x < n16) dst[threadIdx.x] = c.q[threadIdx.x];
}

struct GroupHeader {
  int32_t n, total, atr, btr, f16, prefix_off, args_off, kernel;
};
inline int64_t group_prefix_off() { return (int64_t)sizeof(GroupHeader); }
inline int64_t group_args_off(int n) { return (group_prefix_off() + (int64_t)(n + 1) * 4 + 15) & ~(int64_t)15; }
}  // namespace

extern "C" int64_t ttsk_gemm_group_table_bytes(int n) {   // rounded up to 16 bytes (uploaded in 16-byte pieces)
  return n > 0 ? (group_args_off(n) + (int64_t)n * (int64_t)sizeof(Args) + 15) & ~(int64_t)15 : 0;
}

extern "C" int ttsk_gemm_group_build(const ttsk_gemm_desc* descs, int n, void* host_table, int32_t* total_wgs) {
  TTSK_REQUIRE(descs && host_table && total_wgs && n > 0 && n <= 4096, "ttsk_gemm_group_build: bad arguments");
  unsigned char* base = (unsigned char*)host_table;
  GroupHeader* h = (GroupHeader*)base;
  int32_t* prefix = (int32_t*)(base + group_prefix_off());
  Args* args = (Args*)(base + group_args_off(n));
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    Args g;
    g.d = descs[i];
    ttsk_gemm_desc& d = g.d;
    if (d.kernel == 0) d.kernel = 1;                // a group runs ONE tile configuration: problem 0's (default 128x128)
    TTSK_REQUIRE(d.kernel != 3, "ttsk_gemm_group_build: the 64-row tile (kernel = 3) has no grouped launch");
    TTSK_REQUIRE(!(d.flags & TTSK_GEMM_RAW_SLABS), "ttsk_gemm_group_build: RAW_SLABS problems are launched with ttsk_gemm");
    if (i == 0) h->kernel = d.kernel;
    TTSK_REQUIRE(d.kernel == h->kernel, "ttsk_gemm_group_build: problem %d asks for kernel %d, problem 0 for %d", i, d.kernel, h->kernel);
    const int rc = validate(d);
    if (rc != TTSK_OK) return rc;
    const int atr = (d.flags & TTSK_GEMM_A_TR) ? 1 : 0, btr = (d.flags & TTSK_GEMM_B_TR) ? 1 : 0, f16 = (d.flags & TTSK_GEMM_F16) ? 1 : 0;
    if (i == 0) { h->atr = atr; h->btr = btr; h->f16 = f16; }
    TTSK_REQUIRE(atr == h->atr && btr == h->btr && f16 == h->f16, "ttsk_gemm_group_build: problem %d has another operand layout / dtype than problem 0", i);
    TTSK_REQUIRE(!atr || btr, "ttsk_gemm_group_build: A_TR needs B_TR");
    TTSK_REQUIRE(!(atr && f16), "ttsk_gemm_group_build: no fp16 instance for transposed A");
    Plan p = make_plan(d);
    if (p.splits > 1)
      TTSK_REQUIRE(d.workspace && d.workspace_bytes >= p.ws_bytes && (((uintptr_t)d.workspace) & 15) == 0,
                   "ttsk_gemm_group_build: problem %d needs a 16-byte aligned split-K workspace of %lld bytes", i, (long long)p.ws_bytes);
    d.splits = p.splits;
    g.tiles_m = p.tiles_m; g.tiles_n = p.tiles_n; g.kchunks = p.kchunks; g.chunks_per_split = p.chunks_per_split;
    prefix[i] = (int32_t)total;
    total += (int64_t)p.tiles_m * p.tiles_n * d.nz1 * d.nz2 * p.splits;
    TTSK_REQUIRE(total < ((int64_t)1 << 30), "ttsk_gemm_group_build: too many workgroups");
    args[i] = g;
  }
  prefix[n] = (int32_t)total;
  h->n = n; h->total = (int32_t)total; h->prefix_off = (int32_t)group_prefix_off(); h->args_off = (int32_t)group_args_off(n);
  *total_wgs = (int32_t)total;
  return TTSK_OK;
}

extern "C" int ttsk_gemm_group_launch_capped(const void* host_table, void* dev_table, int max_wgs, void* stream);
extern "C" int ttsk_gemm_group_launch(const void* host_table, void* dev_table, void* stream) {
  return ttsk_gemm_group_launch_capped(host_table, dev_table, 0, stream);
}

static bool group_inline_table(const GroupHeader* h) { return h->n <= 2 && !h->f16 && h->kernel == 1; }      // bf16 pairs: table in the kernel arguments

// the table -> device memory through kernel arguments (capturable); a no-op for a group whose table travels inline
extern "C" int ttsk_gemm_group_upload(const void* host_table, void* dev_table, void* stream) {
  TTSK_REQUIRE(host_table && dev_table && (((uintptr_t)dev_table) & 15) == 0, "ttsk_gemm_group_upload: bad table pointers");
  const GroupHeader* h = (const GroupHeader*)host_table;
  TTSK_REQUIRE(h->n > 0 && h->total > 0, "ttsk_gemm_group_upload: empty table (call ttsk_gemm_group_build first)");
  if (group_inline_table(h)) return TTSK_OK;
  const int64_t bytes = ttsk_gemm_group_table_bytes(h->n);
  for (int64_t off = 0; off < bytes; off += (int64_t)sizeof(TableChunk)) {
    TableChunk c;
    const int64_t nb = bytes - off < (int64_t)sizeof(TableChunk) ? bytes - off : (int64_t)sizeof(TableChunk);
    memset(&c, 0, sizeof(c));
    memcpy(&c, (const unsigned char*)host_table + off, (size_t)nb);
    hipLaunchKernelGGL(group_table_upload_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, c,
                       (uint4*)((unsigned char*)dev_table + off), (int)((nb + 15) / 16));
    TTSK_CHECK_LAUNCH();
  }
  return TTSK_OK;
}

extern "C" int ttsk_gemm_group_launch_uploaded(const void* host_table, void* dev_table, int max_wgs, void* stream);
extern "C" int ttsk_gemm_group_launch_capped(const void* host_table, void* dev_table, int max_wgs, void* stream) {
  if (int rc = ttsk_gemm_group_upload(host_table, dev_table, stream)) return rc;
  return ttsk_gemm_group_launch_uploaded(host_table, dev_table, max_wgs, stream);
}

// the launch alone: dev_table already holds the table (ttsk_gemm_group_upload, possibly on another stream the caller has ordered
// before this one) — the upload's three dependent 5-us launches need not sit between two grouped launches on the same stream
extern "C" int ttsk_gemm_group_launch_uploaded(const void* host_table, void* dev_table, int max_wgs, void* stream) {
  TTSK_REQUIRE(host_table && dev_table && (((uintptr_t)dev_table) & 15) == 0, "ttsk_gemm_group_launch: bad table pointers");
  TTSK_REQUIRE(max_wgs >= 0, "ttsk_gemm_group_launch_capped: max_wgs < 0");
  const GroupHeader* h = (const GroupHeader*)host_table;
  TTSK_REQUIRE(h->n > 0 && h->total > 0, "ttsk_gemm_group_launch: empty table (call ttsk_gemm_group_build first)");
  const bool inline_table = group_inline_table(h);
  const int* prefix = (const int*)((const unsigned char*)dev_table + h->prefix_off);
  const Args* args = (const Args*)((const unsigned char*)dev_table + h->args_off);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(h->total), block(NTHREADS);
  if (h->kernel == 2) {
    ttsk_launch_gemm2_group(prefix, args, h->n, h->total, max_wgs, h->atr, h->btr, h->f16, s);
  } else if (inline_table) {
    GroupInline<2> t;
    const int32_t* hp = (const int32_t*)((const unsigned char*)host_table + h->prefix_off);
    const Args* ha = (const Args*)((const unsigned char*)host_table + h->args_off);
    t.n = h->n;
    for (int i = 0; i <= 2; ++i) t.prefix[i] = i <= h->n ? hp[i] : hp[h->n];
    for (int i = 0; i < 2; ++i) t.args[i] = ha[i < h->n ? i : 0];
    if (h->atr) hipLaunchKernelGGL((gemm_group_inline_kernel<true, true, false, 2>), grid, block, 0, s, t);
    else if (h->btr) hipLaunchKernelGGL((gemm_group_inline_kernel<false, true, false, 2>), grid, block, 0, s, t);
    else hipLaunchKernelGGL((gemm_group_inline_kernel<false, false, false, 2>), grid, block, 0, s, t);
  } else if (h->atr) hipLaunchKernelGGL((gemm_group_kernel<true, true, false>), grid, block, 0, s, prefix, args, h->n);
  else if (h->btr) {
    if (h->f16) hipLaunchKernelGGL((gemm_group_kernel<false, true, true>), grid, block, 0, s, prefix, args, h->n);
    else hipLaunchKernelGGL((gemm_group_kernel<false, true, false>), grid, block, 0, s, prefix, args, h->n);
  } else {
    if (h->f16) hipLaunchKernelGGL((gemm_group_kernel<false, false, true>), grid, block, 0, s, prefix, args, h->n);
    else hipLaunchKernelGGL((gemm_group_kernel<false, false, false>), grid, block, 0, s, prefix, args, h->n);
  }
  TTSK_CHECK_LAUNCH();
  // problems that were split along K and do not defer their reduction: one reducer each, after the grouped grid
  const Args* hargs = (const Args*)((const unsigned char*)host_table + h->args_off);
  for (int i = 0; i < h->n; ++i) {
    const ttsk_gemm_desc& d = hargs[i].d;
    if (d.splits > 1 && !(d.flags & TTSK_GEMM_DEFER_REDUCE)) {
      const int64_t work = (int64_t)d.M * ((d.N + 7) / 8);
      int blocks = (int)((work + 255) / 256);
      if (blocks > 1024) blocks = 1024;
      dim3 rgrid(blocks, d.nz1 * d.nz2);
      if (h->f16) hipLaunchKernelGGL((gemm_reduce_kernel<true>), rgrid, dim3(256), 0, s, hargs[i]);
      else hipLaunchKernelGGL((gemm_reduce_kernel<false>), rgrid, dim3(256), 0, s, hargs[i]);
      TTSK_CHECK_LAUNCH();
    }
  }
  return TTSK_OK;
}
