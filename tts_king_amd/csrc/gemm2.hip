// gemm2.hip — the large-tile configuration of the MFMA contraction kernel (same descriptor, same epilogue as gemm.hip).
//
// Why a second configuration: a 128x128x64 tile moves 32 KiB from L2 per 2.1 MFLOP, which at ~64 B/clk/CU of L1/TA
// bandwidth costs as many cycles as its 128 MFMAs — the 128^2 kernel is load-path bound (measured 1.3-1.7 us per K
// step per workgroup, ~20 % MFMA utilisation).  Here: 256(M) x 128(N) x 64(K) per 512-thread workgroup (8 waves as
// 4x2, each a 64x64 sub-tile of 4x4 v_mfma_f32_16x16x32), 48 KiB per K step for 4.2 MFLOP, and the operand tiles go
// global -> LDS directly (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write pass) into a 3-stage ring so that two
// K steps are always in flight behind a counted s_waitcnt vmcnt(6) and ONE raw s_barrier per K step.
// LDS-DMA writes lane-linear 1 KiB pieces, so the bank-conflict swizzle is applied to the per-lane SOURCE address:
// the LDS image is identical to gemm.hip's (row-major tiles: 16-byte chunk ^= row & 7; contraction-major tiles: the
// ds_read_b64_tr_b16 layout).  Tails and conv zero padding are out-of-range buffer offsets (hardware returns zeros).
#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int NTHREADS = 512;
constexpr int NSTAGE = 3;
constexpr int A_BYTES = BM * BK * 2;            // 32 KiB
constexpr int B_BYTES = BN * BK * 2;            // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 48 KiB
constexpr int SMEM_BYTES = NSTAGE * STAGE_BYTES;  // 144 KiB
constexpr int CS_LD = 132;                      // fp32 epilogue tile (128 rows per pass): 67,584 B
constexpr int OOB = 0x7FFFFFFF;
constexpr int NA = A_BYTES / (NTHREADS * 16);   // 4 LDS-DMA pieces per lane per K step for A
constexpr int NB = B_BYTES / (NTHREADS * 16);   // 2 for B

typedef __attribute__((address_space(3))) void* lds_ptr;

// one LDS-DMA piece: 64 lanes x 16 B from per-lane buffer offsets `voff` to the 1 KiB at `lds` (wave-uniform).
// The offset is made opaque first: otherwise hipcc turns the out-of-range select into exec-masked twin loads, which
// breaks the fixed pieces-per-K-step count the s_waitcnt vmcnt(N) accounting relies on.
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff) {
  asm volatile("" : "+v"(voff));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, voff, 0, 0, 0);
}

template <bool ATR, bool BTR, bool F16>
__global__ __launch_bounds__(NTHREADS, 1) void gemm2_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  const ttsk_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap: consecutive logical tiles share an XCD's L2
    const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y, split = blockIdx.z;
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
  int a_off[NA], a_t[NA], a_k[NA], b_off[NB], b_k[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = i * 8 + wave;                         // piece 0..31
    if (!ATR) {
      const int row = p * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (lane >> 3);
      const int gm = m0 + row;
      a_off[i] = gm < M ? (gm * d.lda + c * 8) * 2 : OOB;
      a_t[i] = conv_a ? gm % d.seg_len : 0;
      a_k[i] = c * 8;
    } else {
      const int h = p >> 4, kr = (p & 15) * 4 + (lane >> 4);
      const int pc = lane & 15;
      const int lp = (pc >> 1) ^ tr_sw(kr);
      const int mcol = m0 + h * 128 + (lp * 2 + (pc & 1)) * 8;
      a_off[i] = mcol < M ? (kr * d.lda + mcol) * 2 : OOB;
      a_t[i] = 0;
      a_k[i] = kr;
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int p = i * 8 + wave;                         // piece 0..15
    if (!BTR) {
      const int row = p * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (lane >> 3);
      const int gn = n0 + row;
      b_off[i] = gn < N ? (gn * d.ldb + c * 8) * 2 : OOB;
      b_k[i] = c * 8;
    } else {
      const int kr = p * 4 + (lane >> 4);
      const int pc = lane & 15;
      const int lp = (pc >> 1) ^ tr_sw(kr);
      const int ncol = n0 + (lp * 2 + (pc & 1)) * 8;
      b_off[i] = ncol < N ? (kr * d.ldb + ncol) * 2 : OOB;
      b_k[i] = kr;
    }
  }

  const int kc_begin = split * g.chunks_per_split;
  int kc_end = kc_begin + g.chunks_per_split;
  if (kc_end > g.kchunks) kc_end = g.kchunks;
  const int per = kc_end > kc_begin ? kc_end - kc_begin : 0;
  const int nk = per * taps;

  auto issue_tile = [&](int kt) __attribute__((always_inline)) {
    unsigned char* sa = smem + (kt % NSTAGE) * STAGE_BYTES;
    unsigned char* sb = sa + A_BYTES;
    const int tap = kt / per;
    const int kbase = (kc_begin + (kt - tap * per)) * BK;
    if (!ATR) {
      const int shift = conv_a ? d.tap_shift0 + tap * d.tap_dshift : 0;
      const int add = (shift * d.lda + kbase) * 2;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int tt = a_t[i] + shift;
        const bool ok = (kbase + a_k[i] < K8) && (!conv_a || (tt >= 0 && tt < d.seg_len)) && a_off[i] != OOB;
        dma16(rsA, sa + (i * 8 + wave) * 1024, ok ? a_off[i] + add : OOB);
      }
    } else {
      const int add = kbase * d.lda * 2;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = (kbase + a_k[i] < K) && a_off[i] != OOB;
        dma16(rsA, sa + (i * 8 + wave) * 1024, ok ? a_off[i] + add : OOB);
      }
    }
    const int tapoff = tap * (int)d.b_tap_stride;
    if (!BTR) {
      const int add = (tapoff + kbase) * 2;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool ok = (kbase + b_k[i] < K8) && b_off[i] != OOB;
        dma16(rsB, sb + (i * 8 + wave) * 1024, ok ? b_off[i] + add : OOB);
      }
    } else {
      const int add = ((kbase + bshift) * d.ldb + tapoff) * 2;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int kk = kbase + b_k[i];
        bool ok = kk < K && b_off[i] != OOB;
        if (d.bseg_len > 0) { const int tt = kk % d.bseg_len + bshift; ok = ok && tt >= 0 && tt < d.bseg_len; }
        dma16(rsB, sb + (i * 8 + wave) * 1024, ok ? b_off[i] + add : OOB);
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lg = lane >> 4;

  auto compute_tile = [&](int kt) __attribute__((always_inline)) {
    const unsigned char* sa = smem + (kt % NSTAGE) * STAGE_BYTES;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!ATR) {
          const int row = wm * 64 + i * 16 + l15;
          af[i] = *(const bf16x8*)(sa + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        } else {
          const unsigned char* sub = sa + (wm >> 1) * 16384;
          const int mblk = ((wm & 1) * 64 + i * 16) >> 4;
          const int k0 = ks * 32 + 8 * lg + (l15 >> 2);
          const int k1 = k0 + 4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sub + k0 * 256 + ((mblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sub + k1 * 256 + ((mblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
          af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
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
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(af[i], bfr[j], acc[i][j]);
    }
  };

  // ---- 3-stage ring: tiles kt+1 and kt+2 are in flight while tile kt is multiplied
  if (nk > 0) {
    issue_tile(0);
    if (nk > 1) issue_tile(1);
    for (int kt = 0; kt < nk; ++kt) {
      // this wave's pieces of tile kt have landed once at most one younger tile (NA+NB pieces) is outstanding
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();     // every wave's pieces landed; every wave is done reading stage (kt-1) % 3
      if (kt + 2 < nk) issue_tile(kt + 2);
      compute_tile(kt);
    }
  }
  __syncthreads();

  // ---- epilogue, 128 rows per pass through an fp32 LDS tile
  float* cs = (float*)smem;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;
  const int64_t roff = z1 * d.sR1 + z2 * d.sR2;
  const int cg = tid & 15;
  const int gn = n0 + cg * 8;
  const int nvalid = (N - gn) < 8 ? (N - gn) : 8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (d.bias && gn < N && e < nvalid && d.splits <= 1) ? d.bias[gn + e] : 0.f;
  float* ws = d.splits > 1 ? (float*)d.workspace + ((int64_t)split * gridDim.y + z) * ((int64_t)M * N) : nullptr;

#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if ((wm >> 1) == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cs[((wm & 1) * 64 + i * 16 + lg * 4 + r) * CS_LD + wn * 64 + j * 16 + l15] = acc[i][j][r];
    }
    __syncthreads();
    if (gn < N) {
      for (int p = 0; p < 4; ++p) {
        const int row = p * 32 + (tid >> 4);
        const int gm = m0 + pass * 128 + row;
        if (gm >= M) continue;
        const f32x4 lo = *(const f32x4*)(cs + row * CS_LD + cg * 8);
        const f32x4 hi = *(const f32x4*)(cs + row * CS_LD + cg * 8 + 4);
        if (ws) {  // split-K: raw partial sums; the reducer applies the epilogue
          float* wp = ws + (int64_t)gm * N + gn;
          if (nvalid == 8 && (N & 3) == 0) {
            *(f32x4*)wp = lo;
            *(f32x4*)(wp + 4) = hi;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (e < nvalid) wp[e] = lo[e]; if (e + 4 < nvalid) wp[e + 4] = hi[e]; }
          }
        } else {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[e + 4] = hi[e]; }
          epilogue_store<F16>(d, coff, roff, gm, gn, nvalid, v, bias, z2);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

int ttsk_launch_gemm2(const GemmArgs& g, bool atr, bool btr, bool f16, hipStream_t s) {
  dim3 grid(g.tiles_m * g.tiles_n, g.d.nz1 * g.d.nz2, g.d.splits), block(NTHREADS);
  if (atr) hipLaunchKernelGGL((gemm2_kernel<true, true, false>), grid, block, 0, s, g);
  else if (btr) {
    if (f16) hipLaunchKernelGGL((gemm2_kernel<false, true, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm2_kernel<false, true, false>), grid, block, 0, s, g);
  } else {
    if (f16) hipLaunchKernelGGL((gemm2_kernel<false, false, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm2_kernel<false, false, false>), grid, block, 0, s, g);
  }
  return 0;
}
