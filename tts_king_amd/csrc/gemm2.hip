// gemm2.hip — the LDS-DMA configurations of the MFMA contraction kernel (same descriptor, same epilogue as gemm.hip).
//
// BM(M) x 128(N) x 64(K) per workgroup of BM/32 waves (2 columns of waves, each wave a 64x64 sub-tile of 4x4
// v_mfma_f32_16x16x32).  Shipped: BM = 256 (8 waves, 3-stage ring of 48 KiB).  The template also builds BM = 128
// (4 waves, 4-stage ring of 32 KiB); measured on the step's small GEMMs it loses to gemm.hip's register-staged 128^2
// kernel, which runs two workgroups per CU (dec w1 fwd 77 vs 59 us, dW shapes equal), so it is not instantiated.
// The operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write pass).  LDS-DMA writes
// lane-linear 1 KiB pieces, so the bank-conflict swizzle is applied to the per-lane SOURCE address: the LDS image is
// identical to gemm.hip's (row-major tiles: 16-byte chunk ^= row & 7; contraction-major tiles: the ds_read_b64_tr_b16
// layout).  Tails and conv zero padding are out-of-range buffer offsets (hardware returns zeros).
#include "gemm_common.h"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int B_BYTES = BN * BK * 2;            // 16 KiB
constexpr int CS_LD = 132;                      // fp32 epilogue tile (128 rows per pass): 67,584 B
constexpr int OOB = 0x7FFFFFFF;
constexpr int NA = 4;                           // LDS-DMA pieces per lane per K step for A (BM*128 B / (BM/32*64 lanes * 16 B))

template <int BM_> struct Cfg {
  static constexpr int BM = BM_;
  static constexpr int NW = BM_ / 32;                       // waves
  static constexpr int NTHREADS = NW * 64;
  static constexpr int NSTAGE = BM_ == 256 ? 3 : 4;
  static constexpr int A_BYTES = BM_ * BK * 2;              // 32 / 16 KiB
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;     // 48 / 32 KiB
  static constexpr int SMEM_BYTES = NSTAGE * STAGE_BYTES;   // 144 / 128 KiB
  static constexpr int NB = B_BYTES / (NTHREADS * 16);      // 2 / 4 pieces per lane per K step for B
  static constexpr int NQ = NA + NB;
};

// one output tile (bid_in of the problem's tiles_m*tiles_n, batch index z of nzgrid, K range `split`)
template <int BM_, bool ATR, bool BTR, bool F16>
__device__ __forceinline__ void gemm2_tile(const GemmArgs& g, int bid_in, int z, int split, int nzgrid, unsigned char* smem) {
  using CF = Cfg<BM_>;
  constexpr int BM = CF::BM, NW = CF::NW, NTHREADS = CF::NTHREADS, NSTAGE = CF::NSTAGE, A_BYTES = CF::A_BYTES;
  constexpr int STAGE_BYTES = CF::STAGE_BYTES, NB = CF::NB, NQ = CF::NQ;
  const ttsk_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = bid_in;
  {  // XCD-aware bijective remap: consecutive logical tiles share an XCD's L2
    const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z1 = z / d.nz2, z2 = z - z1 * d.nz2;

  const bf16_t* __restrict__ A = (const bf16_t*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const bf16_t* __restrict__ B = (const bf16_t*)d.B + z1 * d.sB1 + z2 * d.sB2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7FFFFFF0, 0x00020000);
  const int M = d.M, N = d.N, K = d.K;
  const int taps = d.taps > 0 ? d.taps : 1;
  const bool conv_a = d.taps > 0;
  const int bshift = d.bseg_len > 0 ? d.bshift0 + z2 * d.bdshift : 0;
  const int K8 = (K + 7) & ~7;

  // ---- per-lane source coordinates of its LDS-DMA pieces.  Piece p of an operand = 1 KiB of the LDS image:
  //      row-major tile: 8 rows x 128 B, lane -> (row p*8 + lane/8, physical chunk lane%8, logical chunk ^ (row&7));
  //      contraction-major tile [64 k][128 m] (A: two of them side by side): 4 k-rows x 256 B,
  //      lane -> (k-row p*4 + lane/16, physical 16-B chunk lane%16; 32-B pair index ^ tr_sw(k-row)).
  // Everything that does not change along K is folded into three per-piece values here, so that a K step issues its six
  // pieces with a handful of VALU instructions and no branch (the address/predicate code used to cost a lone workgroup
  // 0.4 us of every 1.0 us K step):  off = byte offset at k = 0, tap 0;  ok = bit t set when the piece's row exists and
  // is inside its utterance under tap t's shift (conv zero padding);  k = the piece's first contraction index.
  int a_off[NA], a_k[NA], b_off[NB], b_k[NB], b_tt[NB];
  unsigned a_ok[NA], b_ok[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = i * NW + wave;                        // piece 0..BM/8-1
    if (!ATR) {
      const int row = p * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (lane >> 3);
      const int gm = m0 + row;
      a_off[i] = (gm * d.lda + c * 8) * 2;
      a_k[i] = c * 8;
      unsigned okm = 0;
      if (gm < M) {
        if (conv_a) {
          const int t = gm % d.seg_len;
          for (int tp = 0; tp < taps; ++tp) {
            const int tt = t + d.tap_shift0 + tp * d.tap_dshift;
            okm |= (tt >= 0 && tt < d.seg_len) ? (1u << tp) : 0u;
          }
        } else okm = 1u;
      }
      a_ok[i] = okm;
    } else {
      const int h = p >> 4, kr = (p & 15) * 4 + (lane >> 4);
      const int pc = lane & 15;
      const int lp = (pc >> 1) ^ tr_sw(kr);
      const int mcol = m0 + h * 128 + (lp * 2 + (pc & 1)) * 8;
      a_off[i] = (kr * d.lda + mcol) * 2;
      a_k[i] = kr;
      a_ok[i] = mcol < M ? 1u : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int p = i * NW + wave;                        // piece 0..15
    if (!BTR) {
      const int row = p * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (lane >> 3);
      const int gn = n0 + row;
      b_off[i] = (gn * d.ldb + c * 8) * 2;
      b_k[i] = c * 8;
      b_ok[i] = gn < N ? 1u : 0u;
    } else {
      const int kr = p * 4 + (lane >> 4);
      const int pc = lane & 15;
      const int lp = (pc >> 1) ^ tr_sw(kr);
      const int ncol = n0 + (lp * 2 + (pc & 1)) * 8;
      b_off[i] = (kr * d.ldb + ncol) * 2;
      b_k[i] = kr;
      b_ok[i] = ncol < N ? 1u : 0u;
    }
    b_tt[i] = 0;
  }

  const int kc_begin = split * g.chunks_per_split;
  int kc_end = kc_begin + g.chunks_per_split;
  if (kc_end > g.kchunks) kc_end = g.kchunks;
  const int per = kc_end > kc_begin ? kc_end - kc_begin : 0;
  const int nk = per * taps;
  const int klimA = ATR ? K : K8, klimB = BTR ? K : K8;

  // LDS byte address of this wave's first piece (wave-uniform: readfirstlane makes that provable for the "s" operand)
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr)smem + wave * 1024);

  // issue state: the next tile to issue is chunk kc_begin + is_kk of tap is_tap, into ring stage is_stage
  int is_tap = 0, is_kk = 0;
  unsigned is_lds = lds_wave;
  const bool bseg = BTR && d.bseg_len > 0;
  auto reset_btt = [&]() __attribute__((always_inline)) {   // position inside its utterance of the piece's k-row (dW of a conv)
    if (bseg) {
#pragma unroll
      for (int i = 0; i < NB; ++i) b_tt[i] = (kc_begin * BK + b_k[i]) % d.bseg_len;
    }
  };
  reset_btt();

  // One K tile is issued as two halves (A pieces 0,1 + B piece 0; A pieces 2,3 + B piece 1) so that the K loop can
  // spread the six LDS-DMA instructions between its MFMAs: the texture addresser takes 16 clk per 1 KiB piece, and a
  // wave that issues its pieces back to back stalls there (measured 550-800 clk of a 2060-clk K step).
  auto issue_piece_a = [&](int i, int addA, int limA) __attribute__((always_inline)) {
    const bool ok = ((a_ok[i] >> is_tap) & 1u) && a_k[i] < limA;
    dma16(rsA, is_lds + i * (NW * 1024), ok ? a_off[i] + addA : OOB);
  };
  auto issue_piece_b = [&](int i, int addB, int limB) __attribute__((always_inline)) {
    bool ok = b_ok[i] && b_k[i] < limB;
    if (BTR) ok = ok && (!bseg || (unsigned)(b_tt[i] + bshift) < (unsigned)d.bseg_len);
    dma16(rsB, is_lds + A_BYTES + i * (NW * 1024), ok ? b_off[i] + addB : OOB);
  };
  auto issue_adds = [&](int& addA, int& addB, int& limA, int& limB) __attribute__((always_inline)) {
    const int kbase = (kc_begin + is_kk) * BK;
    const int tapoff = is_tap * (int)d.b_tap_stride;
    if (!ATR) addA = ((conv_a ? d.tap_shift0 + is_tap * d.tap_dshift : 0) * d.lda + kbase) * 2;
    else addA = kbase * d.lda * 2;
    if (!BTR) addB = (tapoff + kbase) * 2;
    else addB = ((kbase + bshift) * d.ldb + tapoff) * 2;
    limA = klimA - kbase;
    limB = klimB - kbase;
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    is_lds = (is_lds == lds_wave + (NSTAGE - 1) * STAGE_BYTES) ? lds_wave : is_lds + STAGE_BYTES;
    if (++is_kk == per) {
      is_kk = 0;
      ++is_tap;
      reset_btt();
    } else if (bseg) {
      if (d.bseg_len >= BK) {
#pragma unroll
        for (int i = 0; i < NB; ++i) { b_tt[i] += BK; b_tt[i] -= b_tt[i] >= d.bseg_len ? d.bseg_len : 0; }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) b_tt[i] = (b_tt[i] + BK) % d.bseg_len;
      }
    }
  };
  // piece q of the next tile, q = 0..NQ-1 in issue order (A0 A1 B0 A2 A3 B1, or A0 B0 A1 B1 A2 B2 A3 B3 when NB = 4);
  // the tile state advances after the last one
  auto issue_q = [&](int q) __attribute__((always_inline)) {
    int addA, addB, limA, limB;
    issue_adds(addA, addB, limA, limB);
    const bool is_a = NB == 2 ? (q % 3 != 2) : (q % 2 == 0);
    const int idx = NB == 2 ? (is_a ? (q / 3) * 2 + q % 3 : q / 3) : q / 2;
    if (is_a) issue_piece_a(idx, addA, limA);
    else issue_piece_b(idx, addB, limB);
    if (q == NQ - 1) issue_advance();
  };
  // wait until at most y younger tiles (NQ pieces each) of this wave are outstanding
  auto wait_tiles = [&](int y) __attribute__((always_inline)) {
    if (y >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NQ) : "memory");
    else if (y == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ) : "memory");
    else if (y == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lg = lane >> 4;

  // fragment reads of one 32-wide contraction half (ks) of a ring stage
  auto read_a = [&](const unsigned char* sa, int ks, int i) __attribute__((always_inline)) -> bf16x8 {
    if (!ATR) {
      const int row = wm * 64 + i * 16 + l15;
      return *(const bf16x8*)(sa + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
    } else {
      const unsigned char* sub = sa + (wm >> 1) * 16384;
      const int mblk = ((wm & 1) * 64 + i * 16) >> 4;
      const int k0 = ks * 32 + 8 * lg + (l15 >> 2);
      const int k1 = k0 + 4;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) bf16x4*)(sub + k0 * 256 + ((mblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) bf16x4*)(sub + k1 * 256 + ((mblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
      return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto read_b = [&](const unsigned char* sa, int ks, int i) __attribute__((always_inline)) -> bf16x8 {
    const unsigned char* sb = sa + A_BYTES;
    if (!BTR) {
      const int row = wn * 64 + i * 16 + l15;
      return *(const bf16x8*)(sb + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
    } else {
      const int nblk = (wn * 64 + i * 16) >> 4;
      const int k0 = ks * 32 + 8 * lg + (l15 >> 2);
      const int k1 = k0 + 4;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) bf16x4*)(sb + k0 * 256 + ((nblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) bf16x4*)(sb + k1 * 256 + ((nblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
      return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };

  // ---- K loop.  NSTAGE-deep LDS ring filled by LDS-DMA; the fragments of one contraction half are read from LDS while
  // the MFMAs of the previous half run (two register sets), and the DMA pieces of the tile NSTAGE steps ahead are spread
  // between the MFMA groups (the texture addresser takes 16 clk per 1 KiB piece: a wave that issues its pieces back to
  // back stalls there).  One s_barrier per K step, in the middle:
  //   phase A (step kt): read F1 <- stage[kt] half 1 | MFMA F0 | vmcnt, lgkmcnt(0), barrier
  //   phase B          : read F0 <- stage[kt+1] half 0 | MFMA F1 | DMA of tile kt+NSTAGE (into stage[kt], free now)
  // At the barrier every wave's reads of stage[kt] have returned and its pieces of tile kt+1 have landed (the younger
  // DMA are those of tiles kt+2 .. kt+NSTAGE-1).  sched_barrier(0) pins the interleave.
#define SB() __builtin_amdgcn_sched_barrier(0)
  if (nk > 0) {
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int t = 0; t < NSTAGE; ++t) {
      if (t < nk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) issue_q(q);
      }
    }
    wait_tiles((nk < NSTAGE ? nk : NSTAGE) - 1);
    __builtin_amdgcn_s_barrier();
    const unsigned char* st = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0[i] = read_a(st, 0, i); b0[i] = read_b(st, 0, i); }
    constexpr int S0 = (NQ + 2) / 3, S1 = (NQ + 1) / 3;     // DMA pieces per MFMA group: 2,2,2 or 3,3,2
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* st_next = (st == smem + (NSTAGE - 1) * STAGE_BYTES) ? smem : st + STAGE_BYTES;
      const bool more = kt + NSTAGE < nk;
      // ---------------- phase A
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = read_b(st, 1, j);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] = mfma16<F16>(a0[0], b0[j], acc[0][j]);
      SB();
      a1[0] = read_a(st, 1, 0); a1[1] = read_a(st, 1, 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[1][j] = mfma16<F16>(a0[1], b0[j], acc[1][j]);
      SB();
      a1[2] = read_a(st, 1, 2); a1[3] = read_a(st, 1, 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[2][j] = mfma16<F16>(a0[2], b0[j], acc[2][j]);
      SB();
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[3][j] = mfma16<F16>(a0[3], b0[j], acc[3][j]);
      SB();
      {
        const int rem = nk - 2 - kt;
        wait_tiles(rem < 0 ? 0 : (rem > NSTAGE - 2 ? NSTAGE - 2 : rem));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      SB();
      // ---------------- phase B
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = read_b(st_next, 0, j);
      if (more) {
#pragma unroll
        for (int q = 0; q < S0; ++q) issue_q(q);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] = mfma16<F16>(a1[0], b1[j], acc[0][j]);
      SB();
      a0[0] = read_a(st_next, 0, 0); a0[1] = read_a(st_next, 0, 1);
      if (more) {
#pragma unroll
        for (int q = S0; q < S0 + S1; ++q) issue_q(q);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[1][j] = mfma16<F16>(a1[1], b1[j], acc[1][j]);
      SB();
      a0[2] = read_a(st_next, 0, 2); a0[3] = read_a(st_next, 0, 3);
      if (more) {
#pragma unroll
        for (int q = S0 + S1; q < NQ; ++q) issue_q(q);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[2][j] = mfma16<F16>(a1[2], b1[j], acc[2][j]);
      SB();
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[3][j] = mfma16<F16>(a1[3], b1[j], acc[3][j]);
      SB();
      st = st_next;
    }
  }
#undef SB
  __syncthreads();

  tile_epilogue<BM, NTHREADS, F16>(d, smem, acc, m0, n0, z, z1, z2, split, nzgrid, wm, wn, lane, tid);
}

template <int BM_, bool ATR, bool BTR, bool F16>
__global__ __launch_bounds__(Cfg<BM_>::NTHREADS, 1) void gemm2_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg<BM_>::SMEM_BYTES];
  gemm2_tile<BM_, ATR, BTR, F16>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, smem);
}

// grouped launch (see gemm.hip: gemm_group_kernel): the workgroups of n problems as one grid
template <int BM_, bool ATR, bool BTR, bool F16>
__global__ __launch_bounds__(Cfg<BM_>::NTHREADS, 1) void gemm2_group_kernel(const int* __restrict__ prefix, const GemmArgs* __restrict__ args,
                                                                           int n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg<BM_>::SMEM_BYTES];
  // The grid may be smaller than the table's workgroup count (ttsk_gemm_group_launch_capped): a workgroup then walks the tiles
  // blockIdx.x, blockIdx.x + gridDim.x, ...  A capped grid of one workgroup per CU leaves the other CUs to a concurrent stream.
  const int total = prefix[n];
  for (int wg = blockIdx.x; wg < total; wg += gridDim.x) {
    int lo = 0, hi = n;                 // invariant: prefix[lo] <= wg < prefix[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (prefix[mid] <= wg) lo = mid; else hi = mid;
    }
    const int p = __builtin_amdgcn_readfirstlane(lo);
    const GemmArgs& g = args[p];
    const int local = wg - prefix[p];
    const int tiles = g.tiles_m * g.tiles_n, nz = g.d.nz1 * g.d.nz2;
    const int tile = local % tiles, rest = local / tiles;
    gemm2_tile<BM_, ATR, BTR, F16>(g, tile, rest % nz, rest / nz, nz, smem);
    __syncthreads();                    // the epilogue's staging tile is the next tile's operand ring
  }
}

}  // namespace

int ttsk_launch_gemm2_group(const int* prefix, const GemmArgs* args, int n, int total_wgs, int max_wgs, bool atr, bool btr, bool f16,
                            hipStream_t s) {
  constexpr int BM = 256;
  dim3 grid(max_wgs > 0 && max_wgs < total_wgs ? max_wgs : total_wgs), block(Cfg<BM>::NTHREADS);
  if (atr) hipLaunchKernelGGL((gemm2_group_kernel<BM, true, true, false>), grid, block, 0, s, prefix, args, n);
  else if (btr) {
    if (f16) hipLaunchKernelGGL((gemm2_group_kernel<BM, false, true, true>), grid, block, 0, s, prefix, args, n);
    else hipLaunchKernelGGL((gemm2_group_kernel<BM, false, true, false>), grid, block, 0, s, prefix, args, n);
  } else {
    if (f16) hipLaunchKernelGGL((gemm2_group_kernel<BM, false, false, true>), grid, block, 0, s, prefix, args, n);
    else hipLaunchKernelGGL((gemm2_group_kernel<BM, false, false, false>), grid, block, 0, s, prefix, args, n);
  }
  return 0;
}

int ttsk_launch_gemm2(const GemmArgs& g, bool atr, bool btr, bool f16, hipStream_t s) {
  constexpr int BM = 256;
  dim3 grid(g.tiles_m * g.tiles_n, g.d.nz1 * g.d.nz2, g.d.splits), block(Cfg<BM>::NTHREADS);
  if (atr) hipLaunchKernelGGL((gemm2_kernel<BM, true, true, false>), grid, block, 0, s, g);
  else if (btr) {
    if (f16) hipLaunchKernelGGL((gemm2_kernel<BM, false, true, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm2_kernel<BM, false, true, false>), grid, block, 0, s, g);
  } else {
    if (f16) hipLaunchKernelGGL((gemm2_kernel<BM, false, false, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm2_kernel<BM, false, false, false>), grid, block, 0, s, g);
  }
  return 0;
}
