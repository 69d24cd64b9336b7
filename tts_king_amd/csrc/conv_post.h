// conv_post.h — the inner product of HiFi-GAN's conv_post (Conv1d(C -> 1, k), hifi/models.py:198-199) over one 16-byte chunk of a
// channels-last 16-bit row: 8 activations x 8 fp32 weights.  Shared by the streaming kernel (hifigan.hip: conv_post_kernel) and the
// fused last stage (mrf32.hip), which must sum in the same order to stay bit-identical.
#pragma once
#include "common.h"

template <bool F16>
__device__ __forceinline__ float conv_post_dot8(const uint4 v, const float* __restrict__ w) {
  float a0, a1, a2, a3, a4, a5, a6, a7;
  unpack2<F16>(v.x, a0, a1); unpack2<F16>(v.y, a2, a3); unpack2<F16>(v.z, a4, a5); unpack2<F16>(v.w, a6, a7);
  const f32x4 w0 = *(const f32x4*)w, w1 = *(const f32x4*)(w + 4);
  return a0 * w0[0] + a1 * w0[1] + a2 * w0[2] + a3 * w0[3] + a4 * w1[0] + a5 * w1[1] + a6 * w1[2] + a7 * w1[3];
}
