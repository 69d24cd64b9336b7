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
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;     // 32 KiB
constexpr int CS_LD = 132;                               // fp32 epilogue tile leading dim (floats)
constexpr int SMEM_BYTES = BM * CS_LD * 4;               // 67,584 B >= 2 stages (65,536 B)

struct Args {
  ttsk_gemm_desc d;
  int tiles_m, tiles_n, kchunks, chunks_per_split;
};

__device__ __forceinline__ unsigned lrelu2(unsigned w, float sl) {
  float lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xFFFF0000u);
  lo = lo > 0.f ? lo : lo * sl;
  hi = hi > 0.f ? hi : hi * sl;
  return pack_bf2(lo, hi);
}
__device__ __forceinline__ uint4 lrelu8(uint4 v, float sl) {
  return make_uint4(lrelu2(v.x, sl), lrelu2(v.y, sl), lrelu2(v.z, sl), lrelu2(v.w, sl));
}
__device__ __forceinline__ void add_bf8(float v[8], uint4 r) {
  v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xFFFF0000u);
  v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xFFFF0000u);
  v[4] += __uint_as_float(r.z << 16); v[5] += __uint_as_float(r.z & 0xFFFF0000u);
  v[6] += __uint_as_float(r.w << 16); v[7] += __uint_as_float(r.w & 0xFFFF0000u);
}

__device__ __forceinline__ int tr_sw(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

template <bool ATR, bool BTR>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  const ttsk_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile id with XCD-aware (bijective) remap: consecutive logical tiles share an XCD's L2
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y, split = blockIdx.z;
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
  const int nrow = tid >> 3, nslot = tid & 7;      // + 32 i rows
  const int trow = tid >> 4, tslot = tid & 15;     // + 16 i krows
  int a_off[4], b_off[4];                          // byte offsets of this thread's 4 chunks at k = 0, tap 0 (OOB if never valid)
  int a_t[4];                                      // conv-A: position of row inside its segment
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!ATR) {
      const int gm = m0 + nrow + 32 * i;
      a_off[i] = gm < M ? (gm * d.lda + nslot * 8) * 2 : OOB;
      a_t[i] = conv_a ? gm % d.seg_len : 0;
    } else {
      const int mcol = m0 + tslot * 8;
      a_off[i] = mcol < M ? ((trow + 16 * i) * d.lda + mcol) * 2 : OOB;
      a_t[i] = 0;
    }
    if (!BTR) {
      const int gn = n0 + nrow + 32 * i;
      b_off[i] = gn < N ? (gn * d.ldb + nslot * 8) * 2 : OOB;
    } else {
      const int ncol = n0 + tslot * 8;
      b_off[i] = ncol < N ? ((trow + 16 * i) * d.ldb + ncol) * 2 : OOB;
    }
  }

  const int kc_begin = split * g.chunks_per_split;
  int kc_end = kc_begin + g.chunks_per_split;
  if (kc_end > g.kchunks) kc_end = g.kchunks;
  const int per = kc_end > kc_begin ? kc_end - kc_begin : 0;
  const int nk = per * taps;

  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const float in_slope = d.in_slope;
  const bool lrelu_in = d.flags & TTSK_GEMM_LRELU_IN;

#define TTSK_LD(rs, off) __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0))
  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    const int tap = kt / per;
    const int kbase = (kc_begin + (kt - tap * per)) * BK;
    int oa[4], ob[4];
    if (!ATR) {
      const int shift = conv_a ? d.tap_shift0 + tap * d.tap_dshift : 0;
      const int add = (shift * d.lda + kbase) * 2;
      const bool kok = kbase + nslot * 8 < K8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tt = a_t[i] + shift;
        const bool ok = kok && (!conv_a || (tt >= 0 && tt < d.seg_len));
        oa[i] = (ok && a_off[i] != OOB) ? a_off[i] + add : OOB;
      }
    } else {
      const int add = kbase * d.lda * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) oa[i] = (kbase + trow + 16 * i < K && a_off[i] != OOB) ? a_off[i] + add : OOB;
    }
    const int tapoff = tap * (int)d.b_tap_stride;
    if (!BTR) {
      const bool kok = kbase + nslot * 8 < K8;
      const int add = (tapoff + kbase) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) ob[i] = (kok && b_off[i] != OOB) ? b_off[i] + add : OOB;
    } else {
      const int add = ((kbase + bshift) * d.ldb + tapoff) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = kbase + trow + 16 * i;
        bool ok = kk < K && b_off[i] != OOB;
        if (d.bseg_len > 0) { const int tt = kk % d.bseg_len + bshift; ok = ok && tt >= 0 && tt < d.bseg_len; }
        ob[i] = ok ? b_off[i] + add : OOB;
      }
    }
    ra0 = TTSK_LD(rsA, oa[0]); ra1 = TTSK_LD(rsA, oa[1]); ra2 = TTSK_LD(rsA, oa[2]); ra3 = TTSK_LD(rsA, oa[3]);
    rb0 = TTSK_LD(rsB, ob[0]); rb1 = TTSK_LD(rsB, ob[1]); rb2 = TTSK_LD(rsB, ob[2]); rb3 = TTSK_LD(rsB, ob[3]);
  };
#undef TTSK_LD

  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    unsigned char* sa = smem + buf * STAGE_BYTES;
    unsigned char* sb = sa + BM * BK * 2;
    if (lrelu_in && !ATR) { ra0 = lrelu8(ra0, in_slope); ra1 = lrelu8(ra1, in_slope); ra2 = lrelu8(ra2, in_slope); ra3 = lrelu8(ra3, in_slope); }
#define TTSK_ST(i, RA, RB)                                                                                        \
    {                                                                                                             \
      if (!ATR) { const int row = nrow + 32 * i; *(uint4*)(sa + row * 128 + ((nslot ^ (row & 7)) << 4)) = RA; }     \
      else { const int kr = trow + 16 * i; *(uint4*)(sa + kr * 256 + (((tslot >> 1) ^ tr_sw(kr)) << 5) + ((tslot & 1) << 4)) = RA; } \
      if (!BTR) { const int row = nrow + 32 * i; *(uint4*)(sb + row * 128 + ((nslot ^ (row & 7)) << 4)) = RB; }     \
      else { const int kr = trow + 16 * i; *(uint4*)(sb + kr * 256 + (((tslot >> 1) ^ tr_sw(kr)) << 5) + ((tslot & 1) << 4)) = RB; } \
    }
    TTSK_ST(0, ra0, rb0) TTSK_ST(1, ra1, rb1) TTSK_ST(2, ra2, rb2) TTSK_ST(3, ra3, rb3)
#undef TTSK_ST
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lg = lane >> 4;

  auto compute_tile = [&](int buf) {
    const unsigned char* sa = smem + buf * STAGE_BYTES;
    const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!ATR) {
          const int row = wm * 64 + i * 16 + l15;
          af[i] = *(const bf16x8*)(sa + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        } else {
          // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block and receives
          // column (lane & 15) of its 4 rows: element j <- tile[k = 8*lg + j][m = block col]
          const int mblk = (wm * 64 + i * 16) >> 4;  // 32-byte slot index of the 16-column block
          const int k0 = ks * 32 + 8 * lg + (l15 >> 2);
          const int k1 = k0 + 4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sa + k0 * 256 + ((mblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sa + k1 * 256 + ((mblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
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
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) load_tile(kt + 1);
      compute_tile(kt & 1);
      if (more) store_tile((kt + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> full-row 16-byte traffic
  float* cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cs[(wm * 64 + i * 16 + lg * 4 + r) * CS_LD + wn * 64 + j * 16 + l15] = acc[i][j][r];
  __syncthreads();

  const int flags = d.flags;
  const bool c32 = flags & TTSK_GEMM_C_F32;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2 + (d.splits > 1 ? split * d.sCs : 0);
  const int64_t roff = z1 * d.sR1 + z2 * d.sR2;
  const int cg = tid & 15;
  const int gn = n0 + cg * 8;
  if (gn >= N) return;
  const int nvalid = (N - gn) < 8 ? (N - gn) : 8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (d.bias && e < nvalid) ? d.bias[gn + e] : 0.f;

  for (int p = 0; p < 8; ++p) {
    const int row = p * 16 + (tid >> 4);
    const int gm = m0 + row;
    if (gm >= M) continue;
    int64_t orow = gm;
    if (d.out_mul != 0) {
      const int s = gm / d.seg_len, t = gm - s * d.seg_len;
      const int o = t * d.out_mul + d.out_add;
      if (o < 0 || o >= d.out_seg) continue;
      orow = (int64_t)s * d.out_seg + o;
    }
    float v[8];
    {
      const f32x4 lo = *(const f32x4*)(cs + row * CS_LD + cg * 8);
      const f32x4 hi = *(const f32x4*)(cs + row * CS_LD + cg * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[e + 4] = hi[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * d.alpha + bias[e];
    if (flags & TTSK_GEMM_ADD_R) {
      if (flags & TTSK_GEMM_R_F32) {
        const float* rp = (const float*)d.R + roff + orow * d.ldr + gn;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += rp[e];
      } else {
        const bf16_t* rp = (const bf16_t*)d.R + roff + orow * d.ldr + gn;
        if (nvalid == 8 && ((d.ldr & 7) == 0)) {
          add_bf8(v, *(const uint4*)rp);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += bf2f(rp[e]);
        }
      }
    }
    if (flags & TTSK_GEMM_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (flags & TTSK_GEMM_MASK_G) {
      const bf16_t* gp = (const bf16_t*)d.G + roff + orow * d.ldg + gn;
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] = bf2f(gp[e]) > 0.f ? v[e] : 0.f;
    }
    if (flags & TTSK_GEMM_LRELU_OUT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.out_slope;
    }
    if (flags & TTSK_GEMM_TANH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
    }
    if (c32) {
      float* cp = (float*)d.C + coff + orow * d.ldc + gn;
      if (flags & TTSK_GEMM_ACCUM_C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] += v[e];
      } else if (nvalid == 8 && ((d.ldc & 3) == 0) && ((coff & 3) == 0)) {
        *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = v[e];
      }
    } else {
      bf16_t* cp = (bf16_t*)d.C + coff + orow * d.ldc + gn;
      if (nvalid == 8 && ((d.ldc & 7) == 0) && ((coff & 7) == 0)) {
        *(uint4*)cp = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = f2bf(v[e]);
      }
    }
    if (d.C2) {
      bf16_t* cp = (bf16_t*)d.C2 + coff + orow * d.ldc + gn;
      if (nvalid == 8 && ((d.ldc & 7) == 0) && ((coff & 7) == 0)) {
        *(uint4*)cp = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = f2bf(v[e]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs,
                                                           int64_t stride, float* __restrict__ dst, int64_t n4,
                                                           int64_t numel, int accumulate) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    if (e + 4 <= numel) {
      f32x4 s = accumulate ? *(const f32x4*)(dst + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < n_slabs; ++k) s += *(const f32x4*)(slabs + k * stride + e);
      *(f32x4*)(dst + e) = s;
    } else {
      for (int64_t j = e; j < numel; ++j) {
        float s = accumulate ? dst[j] : 0.f;
        for (int k = 0; k < n_slabs; ++k) s += slabs[k * stride + j];
        dst[j] = s;
      }
    }
  }
}

}  // namespace

extern "C" int ttsk_gemm(const ttsk_gemm_desc* dp, void* stream) {
  TTSK_REQUIRE(dp != nullptr, "ttsk_gemm: null descriptor");
  Args g;
  g.d = *dp;
  ttsk_gemm_desc& d = g.d;
  TTSK_REQUIRE(d.A && d.B && d.C, "ttsk_gemm: null operand");
  TTSK_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "ttsk_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  const bool atr = d.flags & TTSK_GEMM_A_TR, btr = d.flags & TTSK_GEMM_B_TR;
  TTSK_REQUIRE(!(atr && !btr), "ttsk_gemm: A_TR without B_TR is not instantiated");
  TTSK_REQUIRE((d.lda & 7) == 0 && (d.ldb & 7) == 0, "ttsk_gemm: lda/ldb must be multiples of 8 (16-byte rows)");
  TTSK_REQUIRE((((uintptr_t)d.A) & 15) == 0 && (((uintptr_t)d.B) & 15) == 0, "ttsk_gemm: A/B must be 16-byte aligned");
  TTSK_REQUIRE(((d.sA1 | d.sA2 | d.sB1 | d.sB2) & 7) == 0, "ttsk_gemm: batch strides of A/B must be multiples of 8");
  TTSK_REQUIRE(d.taps == 0 || (!atr && d.seg_len > 0 && d.M % d.seg_len == 0), "ttsk_gemm: conv-A mode needs A untransposed and M %% seg_len == 0");
  TTSK_REQUIRE(d.taps == 0 || (d.b_tap_stride & 7) == 0, "ttsk_gemm: b_tap_stride must be a multiple of 8");
  TTSK_REQUIRE(d.bseg_len == 0 || btr, "ttsk_gemm: B row shift needs B_TR");
  TTSK_REQUIRE(d.out_mul == 0 || (d.seg_len > 0 && d.out_seg > 0), "ttsk_gemm: output remap needs seg_len/out_seg");
  if (d.nz1 < 1) d.nz1 = 1;
  if (d.nz2 < 1) d.nz2 = 1;
  if (d.splits < 1) d.splits = 1;
  TTSK_REQUIRE(d.splits == 1 || ((d.flags & TTSK_GEMM_C_F32) && d.taps <= 1), "ttsk_gemm: split-K needs fp32 C and no taps");
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_ADD_R) || d.R, "ttsk_gemm: ADD_R without R");
  TTSK_REQUIRE(!(d.flags & TTSK_GEMM_MASK_G) || d.G, "ttsk_gemm: MASK_G without G");
  g.tiles_m = (d.M + BM - 1) / BM;
  g.tiles_n = (d.N + BN - 1) / BN;
  g.kchunks = (d.K + BK - 1) / BK;
  g.chunks_per_split = (g.kchunks + d.splits - 1) / d.splits;
  dim3 grid(g.tiles_m * g.tiles_n, d.nz1 * d.nz2, d.splits), block(NTHREADS);
  TTSK_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "ttsk_gemm: batch/splits too large");
  hipStream_t s = (hipStream_t)stream;
  if (atr)
    hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, s, g);
  else if (btr)
    hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, s, g);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_reduce_slabs(const float* slabs, int n_slabs, int64_t slab_stride, float* dst, int64_t numel,
                                 int accumulate, void* stream) {
  TTSK_REQUIRE(slabs && dst && n_slabs > 0 && numel > 0, "ttsk_reduce_slabs: bad arguments");
  TTSK_REQUIRE((slab_stride & 3) == 0 && (((uintptr_t)slabs | (uintptr_t)dst) & 15) == 0,
               "ttsk_reduce_slabs: slabs/dst must be 16-byte aligned");
  const int64_t n4 = (numel + 3) / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs, slab_stride,
                     dst, n4, numel, accumulate);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
