// Shared device/host helpers for libttsk_hip (gfx950 only: wave = 64 lanes, no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ttsk.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
  return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
}
// Reductions over the 16 lanes that share lane >> 4 (one MFMA output row lives in such a group = one DPP row).  Four DPP steps
// on the VALU — xor 1, xor 2 inside the quad, then row_half_mirror (quads 0<->1, 2<->3) and row_mirror (halves) — instead of
// four ds_bpermute round trips through the LDS (the online softmax runs two of these per row and key tile, back to back on the
// wave's critical path).  max / + are commutative, so the mirrored pairings reduce like the xor butterfly; every lane ends with
// the group's result.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f32<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f32<0x141>(v));     // row_half_mirror
  v = fmaxf(v, dpp_f32<0x140>(v));     // row_mirror
  return v;
}
__device__ __forceinline__ float quad16_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {      // four DPP steps inside the 16-lane rows, two LDS permutes across them
  v = quad16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = quad16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// 16-bit element codecs: bf16 (FS2 training) or IEEE fp16 (HiFi-GAN inference: 3 more mantissa bits at the same MFMA rate)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <bool F16> __device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
  if constexpr (F16) {
    const half2v h = __builtin_bit_cast(half2v, w);
    lo = (float)h[0]; hi = (float)h[1];
  } else {
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xFFFF0000u);
  }
}
template <bool F16> __device__ __forceinline__ unsigned pack2(float lo, float hi) {
  if constexpr (F16) {
    half2v h; h[0] = (_Float16)lo; h[1] = (_Float16)hi;      // v_cvt_f16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, h);
  } else {
    return pack_bf2(lo, hi);
  }
}
template <bool F16> __device__ __forceinline__ float unpack1(bf16_t v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return bf2f(v);
}
template <bool F16> __device__ __forceinline__ bf16_t pack1(float f) {
  if constexpr (F16) return __builtin_bit_cast(bf16_t, (_Float16)f);
  else return f2bf(f);
}
template <bool F16> __device__ __forceinline__ unsigned lrelu2(unsigned w, float sl) {
  float lo, hi;
  unpack2<F16>(w, lo, hi);
  lo = lo > 0.f ? lo : lo * sl;
  hi = hi > 0.f ? hi : hi * sl;
  return pack2<F16>(lo, hi);
}
template <bool F16> __device__ __forceinline__ uint4 lrelu8(uint4 v, float sl) {
  return make_uint4(lrelu2<F16>(v.x, sl), lrelu2<F16>(v.y, sl), lrelu2<F16>(v.z, sl), lrelu2<F16>(v.w, sl));
}
// LeakyReLU of two packed 16-bit values, bit for bit lrelu2<F16> (x > 0 ? x : round16(float(x) * slope)) for 0 < slope < 1.  fp16: float(x) * slope is ONE
// v_fma_mix_f32 per element (the conversion rides in the operand), the select is one packed max — x >= slope * x exactly when x >= 0, also after the
// rounding: 4 VALU instructions per pair instead of 9.  Callers guarantee 0 < slope < 1 (the entry points check it).
template <bool F16> __device__ __forceinline__ unsigned lrelu2_fast(unsigned w, float sl) {
  if constexpr (F16) {
    float lo, hi;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(w), "v"(sl));
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(w), "v"(sl));
    const unsigned p = pack2<true>(lo, hi);
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(p));
    return r;
  } else {
    return lrelu2<false>(w, sl);
  }
}

template <bool F16> __device__ __forceinline__ uint4 lrelu8_fast(uint4 v, float sl) {
  return make_uint4(lrelu2_fast<F16>(v.x, sl), lrelu2_fast<F16>(v.y, sl), lrelu2_fast<F16>(v.z, sl), lrelu2_fast<F16>(v.w, sl));
}
template <bool F16> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Philox4x32-10 counter RNG: dropout masks are regenerated in backward from (seed, site, element index),
// never stored (reference dropout sites: SubLayers.py:62,99; modules.py:286,298; Layers.py:137-141).
struct Philox {
  __device__ static __forceinline__ uint4 gen(uint2 key, uint4 ctr) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      unsigned hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
      unsigned hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
      ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
      key.x += 0x9E3779B9u;
      key.y += 0xBB67AE85u;
    }
    return ctr;
  }
};
// keep-mask bits for 4 consecutive elements starting at element index e4*4 of dropout site `site`
__device__ __forceinline__ uint4 dropout_bits(uint64_t seed, unsigned site, unsigned e4) {
  return Philox::gen(make_uint2((unsigned)seed, (unsigned)(seed >> 32)), make_uint4(e4, site, 0x5eedu, 0u));
}
__device__ __forceinline__ unsigned keep_threshold(float p) {  // keep iff bits >= thr  (P(keep) = 1-p)
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

// host-side error plumbing
void ttsk_set_error(const char* fmt, ...);
#define TTSK_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ttsk_set_error(__VA_ARGS__);              \
      return TTSK_EINVAL;                       \
    }                                           \
  } while (0)
#define TTSK_CHECK_LAUNCH()                                                    \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      ttsk_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return TTSK_ELAUNCH;                                                     \
    }                                                                          \
  } while (0)
