// gemm_common.h — shared by the two tile configurations of the MFMA GEMM (gemm.hip: 128x128x64 / 4 waves, register
// staging; gemm2.hip: 256x128x64 / 8 waves, 3-stage LDS-DMA ring): launch arguments and the fused epilogue.
#pragma once
#include "common.h"

struct GemmArgs {
  ttsk_gemm_desc d;
  int tiles_m, tiles_n, kchunks, chunks_per_split;
};

template <bool F16> __device__ __forceinline__ void add_h8(float v[8], uint4 r) {
  float lo, hi;
  unpack2<F16>(r.x, lo, hi); v[0] += lo; v[1] += hi;
  unpack2<F16>(r.y, lo, hi); v[2] += lo; v[3] += hi;
  unpack2<F16>(r.z, lo, hi); v[4] += lo; v[5] += hi;
  unpack2<F16>(r.w, lo, hi); v[6] += lo; v[7] += hi;
}
template <bool F16> __device__ __forceinline__ uint4 pack8(const float v[8]) {
  return make_uint4(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7]));
}

// Epilogue of one output row segment of 8 columns: v = alpha*acc + bias, residual, ReLU / gate / LeakyReLU / tanh, then the
// store(s).  Shared by the GEMM kernel (splits == 1) and by the split-K reducer.
template <bool F16>
__device__ __forceinline__ void epilogue_store(const ttsk_gemm_desc& d, int64_t coff, int64_t roff, int gm, int gn, int nvalid,
                                               float v[8], const float bias[8], int z2 = 0) {
  const int flags = d.flags;
  int64_t orow = gm;
  if (d.out_mul != 0) {
    const int s = gm / d.seg_len, t = gm - s * d.seg_len;
    const int o = t * d.out_mul + d.out_add + z2 * d.out_add_dz;
    if (o < 0 || o >= d.out_seg) return;
    orow = (int64_t)s * d.out_seg + o;
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
      if (nvalid == 8 && ((d.ldr & 7) == 0) && ((roff & 7) == 0)) {
        add_h8<F16>(v, *(const uint4*)rp);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += unpack1<F16>(rp[e]);
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
    for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] = unpack1<F16>(gp[e]) > 0.f ? v[e] : 0.f;
  }
  if (flags & TTSK_GEMM_LRELU_OUT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.out_slope;
  }
  if (flags & TTSK_GEMM_TANH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
  }
  if (flags & TTSK_GEMM_C_F32) {
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
      *(uint4*)cp = pack8<F16>(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = pack1<F16>(v[e]);
    }
  }
  if (d.C2) {
    if (flags & TTSK_GEMM_C2_LRELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.out_slope;
    }
    bf16_t* cp = (bf16_t*)d.C2 + coff + orow * d.ldc + gn;
    if (nvalid == 8 && ((d.ldc & 7) == 0) && ((coff & 7) == 0)) {
      *(uint4*)cp = pack8<F16>(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = pack1<F16>(v[e]);
    }
  }
}

__device__ __forceinline__ int tr_sw(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// gemm2.hip
int ttsk_launch_gemm2(const GemmArgs& g, bool atr, bool btr, bool f16, hipStream_t s);
